#!/bin/bash
# round 6, visit B: the band whose waves expand the j-tile themselves (no LDS image, no barrier) against the committed band
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "area or fuzz or config2" > gpurun_out/r6b_pytest.log 2>&1; rc=$?
echo "[pytest area/fuzz] exit $rc: $(tail -1 gpurun_out/r6b_pytest.log)"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
if [ $rc -ne 0 ]; then tail -30 gpurun_out/r6b_pytest.log; fi
LIBS="libldx_base libldx" AREA=1 ROUNDS=3 SHAPES="3000 5008 fp4 50 k16" bash tools/gpu_abx.sh > gpurun_out/r6b_band_ab.log 2>&1
grep "area2" gpurun_out/r6b_band_ab.log | cut -c1-300
LIB=libldx_ts bash tools/gpu_area_stamps.sh > gpurun_out/r6b_band_selfb_stamps.log 2>&1
grep -v "^  File\|^Traceback\|^    \|LdxError" gpurun_out/r6b_band_selfb_stamps.log | head -12
