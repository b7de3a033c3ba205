#!/bin/bash
# timings of the triangle kernels at the three judged shapes (after gpu_quick.sh's parity tests): FP4 (product) in both cell
# formats, then the int8 kernel (its epilogue is the fp64 tier: what edge units of the FP4 kernel run)
set -u
mkdir -p gpurun_out
for cfg in "10000 5008 fp4 200 k16" "40000 5008 fp4 10 k16" "50000 1008 fp4 10 k16" "10000 5008 fp4 100 ld32" "10000 5008 mfma 50 k16" "50000 1008 mfma 5 k16" ${EXTRA:-}; do
  timeout -k 10 200 python tools/gpu_tri.py $cfg 2>&1 | grep -v amdgpu.ids
done
