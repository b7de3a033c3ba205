#!/bin/bash
# round 6, visit D: long fuzz of the final library (two seeds) + repeated launches at two bench sizes
set -u
mkdir -p gpurun_out
timeout -k 10 420 python tools/gpu_fuzz.py 330 6101 > gpurun_out/r6d_fuzz_a.log 2>&1; echo "[fuzz a] exit $?: $(tail -1 gpurun_out/r6d_fuzz_a.log | cut -c1-200)"
timeout -k 10 420 python tools/gpu_fuzz.py 330 6102 > gpurun_out/r6d_fuzz_b.log 2>&1; echo "[fuzz b] exit $?: $(tail -1 gpurun_out/r6d_fuzz_b.log | cut -c1-200)"
timeout -k 10 300 python tools/gpu_soak.py 50000 1008 200 > gpurun_out/r6d_soak_a.log 2>&1; echo "[soak 50000x1008] exit $?: $(tail -1 gpurun_out/r6d_soak_a.log | cut -c1-200)"
timeout -k 10 300 python tools/gpu_soak.py 10000 5008 500 > gpurun_out/r6d_soak_b.log 2>&1; echo "[soak 10000x5008] exit $?: $(tail -1 gpurun_out/r6d_soak_b.log | cut -c1-200)"
