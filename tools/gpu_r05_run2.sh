#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 300 python tools/gpu_streams.py 10000 5008 200 3 > gpurun_out/streams_10k_fixed.log 2>&1; echo "[streams 10k] $?"; grep -v amdgpu.ids gpurun_out/streams_10k_fixed.log | tail -8
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "[pytest] $?"; tail -8 gpurun_out/pytest_gpu.log
