#!/bin/bash
# timings of the tuning build under LDX_ABLATE values: gpu_ablate.sh "0 16 ..." [snps] [reps]
set -u
timeout -k 10 600 python -c "import torch"
for a in $1; do
  echo -n "ablate=$a  "
  LDX_LIB=$PWD/ld_tools_amd/libldx_tune.so LDX_ABLATE=$a timeout -k 10 200 python tools/gpu_tri.py ${2:-40000} 5008 mfma ${3:-10} 2>&1 | grep -v amdgpu.ids | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms'],4))"
done
