#!/bin/bash
# round 6, visit C: the band's decoded ticket order against the committed band; counters of the self-expanding build
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "area or fuzz or config2" > gpurun_out/r6c_pytest.log 2>&1; rc=$?
echo "[pytest area/fuzz] exit $rc: $(tail -1 gpurun_out/r6c_pytest.log)"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
if [ $rc -ne 0 ]; then tail -30 gpurun_out/r6c_pytest.log; fi
LIBS="libldx_base libldx" AREA=1 ROUNDS=4 SHAPES="3000 5008 fp4 50 k16" bash tools/gpu_abx.sh > gpurun_out/r6c_band_ab.log 2>&1
grep "area2" gpurun_out/r6c_band_ab.log | cut -c1-200
for L in libldx_base libldx_selfb; do
  LDX_LIB=$PWD/ld_tools_amd/$L.so PROG="tools/gpu_exp.py area" PASSES="sq sq3" bash tools/gpu_prof.sh r06d_$L > gpurun_out/r6c_prof_$L.log 2>&1
  echo "[pmc $L] $(tail -1 gpurun_out/r6c_prof_$L.log | cut -c1-200)"
  python3 tools/pmc_table.py gpurun_out/prof_r06d_$L triangle_mfma_kernel > gpurun_out/r6c_cycle_accounting_$L.txt 2>&1
done
