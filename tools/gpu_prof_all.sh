#!/bin/bash
# The round's judged profiles in one visit, as the LAST GPU action of a round:
#   <R>a  bench default workload (configs[1], 10 000 x 5008), the headline line
#   <R>b  40 000 x 5008 (steady state)
#   <R>c  configs[4], 50 000 x 1008, --no-extra-legs --no-other-workloads (a CLEAN kernel average: no two-stream leg)
#   <R>d  configs[2], ld_area 100 000 SNPs (tools/gpu_exp.py area)
# then a default `python bench.py` (no profiler) whose line is kept beside them.
# Usage on the GPU box: bash tools/gpu_prof_all.sh r03      Afterwards, in the build container, right away (the commit must
# be the one that was sent):  python tools/save_profile_all.py r03
# Round 5: seven passes per profile (trace, fetch, write, four SQ counter sets) no longer fit one 20-minute visit:
#   ONLY="a b" bash tools/gpu_prof_all.sh r05      then      ONLY="c d bench" bash tools/gpu_prof_all.sh r05
set -u
R=${1:-r03}
ONLY=${ONLY:-a b c d bench}
COMMON="--no-cpu-baseline --no-other-workloads"
mkdir -p gpurun_out
for what in $ONLY; do
  case $what in
    # (--no-extra-legs since round 5: the two-stream leg's 1 200 OVERLAPPED launches of the same kernel would sit in the
    # kernel-stats average; the legs are on the default bench line that is kept beside the profile)
    a) bash tools/gpu_prof.sh ${R}a --steps 20 --warmup 3 --no-extra-legs $COMMON > gpurun_out/prof_all_${R}a.log 2>&1; echo "[${R}a] exit $?";;
    b) bash tools/gpu_prof.sh ${R}b --steps 10 --warmup 2 --snps 40000 --no-extra-legs $COMMON > gpurun_out/prof_all_${R}b.log 2>&1; echo "[${R}b] exit $?";;
    c) bash tools/gpu_prof.sh ${R}c --steps 10 --warmup 2 --snps 50000 --haps 1008 --no-extra-legs $COMMON > gpurun_out/prof_all_${R}c.log 2>&1; echo "[${R}c] exit $?";;
    d) PROG="tools/gpu_exp.py area" bash tools/gpu_prof.sh ${R}d > gpurun_out/prof_all_${R}d.log 2>&1; echo "[${R}d] exit $?";;
    bench) timeout -k 10 500 python3 bench.py > gpurun_out/bench_default_${R}.json 2> gpurun_out/bench_default_${R}.err; echo "[bench default] exit $?"
           tail -c 400 gpurun_out/bench_default_${R}.json;;
  esac
done
