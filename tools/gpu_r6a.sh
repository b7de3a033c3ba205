#!/bin/bash
# round 6, visit A: the barrier-free band against the committed one (same box, interleaved), and the fp32 tier's new SNP
# classes against rounds 1-5's rule on the two odd panels
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "area or fuzz or config2" > gpurun_out/r6a_pytest.log 2>&1; rc=$?
echo "[pytest area/fuzz] exit $rc: $(tail -1 gpurun_out/r6a_pytest.log)"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
LIBS="libldx_base libldx" AREA=1 ROUNDS=3 SHAPES="10000 5008 fp4 200 k16" bash tools/gpu_abx.sh > gpurun_out/r6a_band_ab.log 2>&1
grep -v "^torch" gpurun_out/r6a_band_ab.log | cut -c1-300
SYNTH_MONO=0.3 LIBS="libldx_r5class libldx" ROUNDS=2 SHAPES="50000 1008 fp4 10 k16" bash tools/gpu_abx.sh > gpurun_out/r6a_mono_ab.log 2>&1
grep -v "^torch" gpurun_out/r6a_mono_ab.log | cut -c1-300
SYNTH_MISS=0.001 SYNTH_MISS_ROWS=0.2 LIBS="libldx_r5class libldx" ROUNDS=2 SHAPES="40000 5008 fp4 10 k16" bash tools/gpu_abx.sh > gpurun_out/r6a_miss_ab.log 2>&1
grep -v "^torch" gpurun_out/r6a_miss_ab.log | cut -c1-300
