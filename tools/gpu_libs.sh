#!/bin/bash
# Time ld_triangle (MFMA path) for several library builds.  usage: gpu_libs.sh "libldx libldx_w8" [snps] [haps] [reps]
# env ABL="13 141": LDX_ABLATE values to run for each library (tuning builds only)
set -u
N=${2:-40000}; H=${3:-5008}; R=${4:-5}
for v in $1; do
  for a in ${ABL:-0}; do
    echo -n "lib=$v ablate=$a "
    LDX_ABLATE=$a LDX_LIB=$PWD/ld_tools_amd/$v.so timeout -k 10 200 python tools/gpu_tri.py $N $H mfma $R 2>&1 | grep -v amdgpu.ids
  done
done
