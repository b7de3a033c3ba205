#!/bin/bash
# does the per-step time depend on how the steps are launched (graph / eager) or on how long the chip has been busy?
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -c "import torch; print('torch', torch.__version__, torch.cuda.is_available())" 2>&1 | tail -1
for args in "--steps 20 --warmup 3" "--no-graph --steps 20 --warmup 3" "--steps 200 --warmup 10" "--steps 5 --warmup 1" "--steps 20 --warmup 3 --settle-steps 0"; do
  echo "== $args"
  timeout -k 10 300 python bench.py $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['value'])"
done
