#!/bin/bash
# In-kernel stamp summary per library build and ablation.  usage: gpu_stamps.sh "libs" "ablates" [snps] [haps]
set -u
N=${3:-40000}; H=${4:-5008}
for v in $1; do for a in $2; do
  echo "== lib=$v ablate=$a"
  F=$PWD/gpurun_out/st_tmp.bin
  LDX_STAMPS=$F LDX_ABLATE=$a LDX_LIB=$PWD/ld_tools_amd/$v.so timeout -k 10 200 python tools/gpu_tri.py $N $H ${P:-fp4} 2 ${FMT:-k16} 2>&1 | grep -v amdgpu.ids
  python tools/stamps.py $F; rm -f $F
done; done
