#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "half_height or triangle_matches_golden or edge_shapes or agree_bitwise or random_shapes or fuzz or sharded_units" > gpurun_out/quart_pytest.log 2>&1; echo "[pytest] $?"; tail -5 gpurun_out/quart_pytest.log
for n in 10000 8000 6000; do echo "== $n"; timeout -k 10 400 python tools/gpu_short.py $n 5008 200 2>&1 | grep -v amdgpu.ids; done
