#!/bin/bash
# Same-box A/B of library builds at the judged shapes, interleaved and repeated (boxes of the pool differ by up to 8 %, and
# so do the first and the tenth second on one box): LIBS="libldx_base libldx" ROUNDS=3 bash tools/gpu_ab2.sh
set -u
mkdir -p gpurun_out
SHAPES=${SHAPES:-"10000 5008 fp4 200 k16|40000 5008 fp4 10 k16|50000 1008 fp4 10 k16"}
for r in $(seq 1 ${ROUNDS:-3}); do
  IFS='|' read -ra S <<< "$SHAPES"
  for shape in "${S[@]}"; do
    for v in ${LIBS:-libldx_base libldx}; do
      echo -n "round=$r lib=$v "
      LDX_LIB=$PWD/ld_tools_amd/$v.so timeout -k 10 200 python tools/gpu_tri.py $shape 2>&1 | grep -v amdgpu.ids
    done
  done
done
