#!/bin/bash
set -u
mkdir -p gpurun_out
run() { local name=$1 tmo=$2; shift 2; timeout -k 10 "$tmo" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?; echo "[$name] exit $rc"; tail -2 "gpurun_out/$name.log" | cut -c1-1500; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
run exp_counts 200 python tools/gpu_exp.py counts
run exp_pack 200 python tools/gpu_exp.py pack
run exp_tri100k 300 python tools/gpu_exp.py tri100k
run exp_eur 300 python tools/gpu_exp.py eur
run exp_area 400 python tools/gpu_exp.py area
run dist_nccl1 300 python bench.py --steps 5 --warmup 2 --force-dist --no-cpu-baseline
run dist_gloo2 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo
