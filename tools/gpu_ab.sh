#!/bin/bash
# A/B timing of tuning builds: for each library in $LIBS (paths under ld_tools_amd/), parity subset then timings
set -u
mkdir -p gpurun_out
# a fresh box pages the image in on the first import (minutes): do that once, outside the timed steps' limits
timeout -k 10 600 python -c "import torch; print('torch', torch.__version__, torch.cuda.is_available())" 2>&1 | tail -1
for lib in ${LIBS:-libldx.so}; do
  echo "== $lib"
  export LDX_LIB=$PWD/ld_tools_amd/$lib
  if [ "${CHECK:-1}" = 1 ]; then
    timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "triangle and not 100k" > gpurun_out/ab_pytest_$lib.log 2>&1; rc=$?
    tail -2 gpurun_out/ab_pytest_$lib.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  fi
  for cfg in "10000 5008 mfma 30" "40000 5008 mfma 3" "50000 1008 mfma 3"; do
    timeout -k 10 300 python tools/gpu_tri.py $cfg 2>&1 | grep -v amdgpu.ids; rc=${PIPESTATUS[0]}
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  done
done
