#!/bin/bash
# quick check after a kernel edit: parity tests of the triangle kernels, then timings
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "triangle or agree" > gpurun_out/quick_pytest.log 2>&1; rc=$?
tail -2 gpurun_out/quick_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for cfg in "10000 5008 mfma 20" "40000 5008 mfma 3" "50000 1008 mfma 3" ${EXTRA:-}; do
  timeout -k 10 200 python tools/gpu_tri.py $cfg 2>&1 | grep -v amdgpu.ids; rc=$?
done
