#!/bin/bash
# eight-wave (NT = 2) FP4 triangle kernel against the four-wave one: parity tests with LDX_QUAD=1, then interleaved timings
set -u
mkdir -p gpurun_out
L=${QLIB:-libldx_q}
LDX_QUAD=1 LDX_LIB=$PWD/ld_tools_amd/$L.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "${K:-triangle or agree}" > gpurun_out/quad_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/quad_pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; [ "${FORCE:-0}" = 1 ] || exit $rc; fi
SHAPES=${SHAPES:-"10000 5008 fp4 200 k16|40000 5008 fp4 10 k16|50000 1008 fp4 10 k16"}
for r in $(seq 1 ${ROUNDS:-2}); do
  IFS='|' read -ra S <<< "$SHAPES"
  for shape in "${S[@]}"; do
    for q in 0 1; do
      echo -n "round=$r quad=$q "
      LDX_QUAD=$q LDX_LIB=$PWD/ld_tools_amd/$L.so timeout -k 10 200 python tools/gpu_tri.py $shape 2>&1 | grep -v amdgpu.ids
    done
  done
done
