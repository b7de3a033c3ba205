#!/bin/bash
set -u
mkdir -p gpurun_out
LIBS="libldx_ff libldx" CHECK="area" AREA=1 ROUNDS=3 SHAPES="10000 5008 fp4 100 k16" bash tools/gpu_abx.sh
for v in libldx_ff libldx; do
  LDX_LIB=$PWD/ld_tools_amd/$v.so PROG="tools/gpu_exp.py area" PASSES="trace fetch write" bash tools/gpu_prof.sh r05d_$v > gpurun_out/prof_band_$v.log 2>&1
  echo "== $v"; grep -A3 "== traffic" gpurun_out/prof_band_$v.log | tail -2; grep "triangle_mfma_kernel<false, false, true" gpurun_out/prof_band_$v.log | head -3
done
