#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { local name=$1 tmo=$2; shift 2; timeout -k 10 "$tmo" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?; echo "[$name] exit $rc"; grep -v amdgpu.ids "gpurun_out/$name.log" | tail -${TAILN:-1} | cut -c1-600; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
run t_mfma_10k 200 python tools/gpu_tri.py 10000 5008 mfma 20
run t_pop_10k 200 python tools/gpu_tri.py 10000 5008 popcount 20
run t_mfma_40k 200 python tools/gpu_tri.py 40000 5008 mfma 3
run t_pop_40k 200 python tools/gpu_tri.py 40000 5008 popcount 3
run t_mfma_eur 200 python tools/gpu_tri.py 50000 1008 mfma 3
run t_pop_eur 200 python tools/gpu_tri.py 50000 1008 popcount 3
OUT=$PWD/gpurun_out/pmc_mfma
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d "$OUT/a" -- python3 tools/gpu_tri.py 40000 5008 mfma 2 > "$OUT/a.log" 2>&1
echo "[pmc a] $?"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/b" -- python3 tools/gpu_tri.py 40000 5008 mfma 2 > "$OUT/b.log" 2>&1
echo "[pmc b] $?"
python3 - <<'PY'
import csv, glob, collections
for tag in ("a", "b"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_mfma/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "triangle" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} n={len(v)} mean={sum(v)/len(v):.5g}")
PY
