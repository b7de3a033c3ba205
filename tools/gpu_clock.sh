#!/bin/bash
# effective shader clock of the triangle kernels: GRBM_GUI_ACTIVE / 8 XCDs / kernel time (MI355X_MICROARCH.md, DVFS)
set -u
export TMPDIR=/tmp
for cfg in "mfma 0" "mfma 5" "mfma 2" "popcount 0"; do
  set -- $cfg
  OUT=$PWD/gpurun_out/clk_$1_$2; rm -rf "$OUT"; mkdir -p "$OUT"
  LDX_ABLATE=$2 timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT" -- python3 tools/gpu_tri.py 40000 5008 $1 3 > "$OUT/log.txt" 2>&1
  python3 - "$OUT" "$1 ablate=$2" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list); dur = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "triangle" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
d = sum(dur) / len(dur)
clk = m["GRBM_GUI_ACTIVE"] / 8 / (d * 1e-9)
simd_cycles = clk * d * 1e-9 * 1024
print(sys.argv[2], f"kernel_ns={d:.0f} clock_GHz={clk/1e9:.3f} mfma_busy={m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/simd_cycles:.3f} "
      f"valu_active={m.get('SQ_ACTIVE_INST_VALU',0)*4/simd_cycles:.3f} wait_any={m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.3f} "
      f"wait_inst={m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.3f} waves_per_simd={m['SQ_WAVE_CYCLES']*4/simd_cycles:.2f}")
PY
done
