#!/bin/bash
set -u
mkdir -p gpurun_out
LIBS="libldx_base libldx_r5b libldx libldx_u4 libldx_u16" ROUNDS=3 SHAPES="10000 5008 fp4 200 k16|3000 5008 fp4 400 k16|40000 5008 fp4 10 k16|50000 1008 fp4 10 k16" PMC=1 bash tools/gpu_abx.sh > gpurun_out/abx_unroll.log 2>&1; grep -v "^\[pmc" gpurun_out/abx_unroll.log | tail -80
