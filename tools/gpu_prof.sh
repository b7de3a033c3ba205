#!/bin/bash
# rocprofv3 evidence for one command: kernel trace + stats, then separate PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"; never --pmc with a trace;
# at most 8 SQ counters per pass; the program itself -- python3 -- directly behind `--`).
# Usage: bash tools/gpu_prof.sh <tag> [bench args...]            -> python3 bench.py <args>
#        PROG="tools/gpu_exp.py area" bash tools/gpu_prof.sh <tag>   -> python3 tools/gpu_exp.py area
#        PASSES="trace sq sq2" bash tools/gpu_prof.sh <tag> ...      -> only these passes (default: all)
#        LDX_LIB=$PWD/ld_tools_amd/libldx_x.so bash tools/gpu_prof.sh <tag> ...   -> another build of the library
# Passes: trace | fetch | write | sq (round 2's eight) | sq2 (instruction mix by class + active cycles by class) |
#         sq3 (co-execution, memory-instruction cycles, LDS issue stalls, busy CUs, FP4 MFMA ops) | sq4 (scalar cycles,
#         branches, instruction fetch, LDS conflicts)  -- round 5, VERDICT r04 item 2: what the SIMD does while neither pipe
#         is busy.  A counter the device refuses fails its pass; the log says which (bench_pmc_<pass>.log).
set -u
TAG=${1:-r02}; shift || true
ARGS=${*:---steps 20 --warmup 3 --no-cpu-baseline}
CMD=${PROG:-bench.py $ARGS}
PASSES=${PASSES:-trace fetch write sq sq2 sq3 sq4}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$PWD"
echo "$CMD" > "$OUT/command.txt"
declare -A PMC
PMC[fetch]="FETCH_SIZE"
PMC[write]="WRITE_SIZE"
PMC[sq]="SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
PMC[sq2]="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
PMC[sq3]="SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM"
PMC[sq4]="SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU"
for p in $PASSES; do
  if [ "$p" = trace ]; then
    timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $CMD > "$OUT/bench_trace.log" 2>&1
    echo "[trace] exit $?"
  else
    timeout -k 10 500 rocprofv3 --pmc ${PMC[$p]} --output-format csv -d "$OUT/pmc_$p" -- python3 $CMD > "$OUT/bench_pmc_$p.log" 2>&1
    echo "[pmc $p] exit $?"
  fi
done
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
