#!/bin/bash
# rocprofv3 evidence for one command: kernel trace + stats, then three separate PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"; never --pmc with a trace).
# Usage: bash tools/gpu_prof.sh <tag> [bench args...]            -> python3 bench.py <args>
#        PROG="tools/gpu_exp.py area" bash tools/gpu_prof.sh <tag>   -> python3 tools/gpu_exp.py area
set -u
TAG=${1:-r02}; shift || true
ARGS=${*:---steps 20 --warmup 3 --no-cpu-baseline}
CMD=${PROG:-bench.py $ARGS}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$PWD"
echo "$CMD" > "$OUT/command.txt"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $CMD > "$OUT/bench_trace.log" 2>&1
echo "[trace] exit $?"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $CMD > "$OUT/bench_pmc_fetch.log" 2>&1
echo "[pmc fetch] exit $?"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $CMD > "$OUT/bench_pmc_write.log" 2>&1
echo "[pmc write] exit $?"
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 $CMD > "$OUT/bench_pmc_sq.log" 2>&1
echo "[pmc sq] exit $?"
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
