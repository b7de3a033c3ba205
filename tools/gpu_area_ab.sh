#!/bin/bash
# ld_area after a band-kernel edit: the area parity tests on the shipped library, then configs[2] timings, A/B interleaved
# (LIBS="libldx_base libldx"), 20 repetitions of the product call each
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "area" > gpurun_out/area_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/area_pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in ${LIBS:-libldx_base libldx}; do
    echo -n "round=$r lib=$v "
    LDX_LIB=$PWD/ld_tools_amd/$v.so timeout -k 10 300 python tools/gpu_exp.py area2 2>&1 | grep -v amdgpu.ids
  done
done
