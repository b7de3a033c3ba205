#!/bin/bash
# A/B of tuning builds under LDX_ABLATE values: usage LIBS="libldx_base libldx_x" ABL="0 1024 1028" bash tools/gpu_abl.sh "50000 1008" "40000 5008"
set -u
for shape in "$@"; do
  for v in ${LIBS:-libldx_x}; do
    for a in ${ABL:-0}; do
      echo -n "lib=$v ablate=$a "
      LDX_ABLATE=$a LDX_LIB=$PWD/ld_tools_amd/$v.so timeout -k 10 200 python tools/gpu_tri.py $shape fp4 ${REPS:-5} k16 2>&1 | grep -v amdgpu.ids
    done
  done
done
