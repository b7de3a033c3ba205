#!/bin/bash
# round 6, last visit: long fuzz of the committed head (two more seeds)
set -u
mkdir -p gpurun_out
timeout -k 10 560 python tools/gpu_fuzz.py 500 6104 > gpurun_out/r6h_fuzz_d.log 2>&1; echo "[fuzz d] exit $?: $(tail -1 gpurun_out/r6h_fuzz_d.log | cut -c1-200)"
timeout -k 10 560 python tools/gpu_fuzz.py 500 6105 > gpurun_out/r6h_fuzz_e.log 2>&1; echo "[fuzz e] exit $?: $(tail -1 gpurun_out/r6h_fuzz_e.log | cut -c1-200)"
