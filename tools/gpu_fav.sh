#!/bin/bash
set -u
timeout -k 10 600 python -c "import torch" 
for N in 10000 40000; do
  F=$PWD/gpurun_out/st_$N.bin
  LDX_STAMPS=$F LDX_LIB=$PWD/ld_tools_amd/libldx_tune.so timeout -k 10 200 python tools/gpu_tri.py $N 5008 mfma 2 2>&1 | grep -v amdgpu.ids
  python tools/stamps_fav.py $F
  python tools/stamps.py $F | grep -i "lifetime\|kernel span\|passes per wave"
done
