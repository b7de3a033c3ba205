#!/bin/bash
# One GPU-box visit: parity tests, smoke, short bench.  A step that is KILLED (timeout) ends the visit;
# a step that merely fails (non-zero exit) is recorded and the next one still runs.
set -u
mkdir -p gpurun_out
run() {   # name, timeout, command...
  local name=$1 tmo=$2; shift 2
  timeout -k 10 "$tmo" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "[$name] exit $rc" | tee -a gpurun_out/summary.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping" | tee -a gpurun_out/summary.log; exit $rc; fi
  return 0
}
: > gpurun_out/summary.log
run pytest_gpu 700 python -m pytest tests -m gpu -q
tail -15 gpurun_out/pytest_gpu.log
run smoke 200 python -c "import __graft_entry__ as g; g.smoke()"
tail -3 gpurun_out/smoke.log
run bench 400 python bench.py --steps 20 --warmup 3
tail -2 gpurun_out/bench.log
