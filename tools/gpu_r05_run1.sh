#!/bin/bash
# round 5, first visit: extended PMC passes on the r04 kernel (two shapes) + the one-stream / two-stream comparison
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -c "import torch; print('torch', torch.__version__, torch.cuda.is_available())" 2>&1 | tail -1
C="--no-cpu-baseline --no-other-workloads --no-extra-legs"
PASSES="sq sq2 sq3 sq4" bash tools/gpu_prof.sh r05pre_a --steps 20 --warmup 3 $C > gpurun_out/r05pre_a.log 2>&1; echo "[pre a] $?"
PASSES="sq sq2 sq3 sq4" bash tools/gpu_prof.sh r05pre_c --steps 10 --warmup 2 --snps 50000 --haps 1008 $C > gpurun_out/r05pre_c.log 2>&1; echo "[pre c] $?"
timeout -k 10 300 python tools/gpu_streams.py 10000 5008 200 5 > gpurun_out/streams_10k.log 2>&1; echo "[streams 10k] $?"; tail -1 gpurun_out/streams_10k.log
timeout -k 10 300 python tools/gpu_streams.py 3000 5008 400 5 > gpurun_out/streams_3k.log 2>&1; echo "[streams 3k] $?"; tail -1 gpurun_out/streams_3k.log
timeout -k 10 300 python tools/gpu_streams.py 40000 5008 10 3 > gpurun_out/streams_40k.log 2>&1; echo "[streams 40k] $?"; tail -1 gpurun_out/streams_40k.log
