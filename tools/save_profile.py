#!/usr/bin/env python3
"""Copy the judged evidence of one tools/gpu_prof.sh run from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/save_profile.py <tag> <round-dir> <workload> [path] [gpus]     e.g.  r01b r01 "ld_triangle 10000x5008"
"""
import glob
import json
import shutil
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
tag, rnd, workload = sys.argv[1], sys.argv[2], sys.argv[3]
path = sys.argv[4] if len(sys.argv) > 4 else "mfma"
gpus = int(sys.argv[5]) if len(sys.argv) > 5 else 1
src = root / "gpurun_out" / f"prof_{tag}"
dst = root / "profiles" / rnd
dst.mkdir(parents=True, exist_ok=True)
shutil.copy(src / "summary.txt", dst / f"{tag}_rocprofv3_summary.txt")
for f in glob.glob(str(src / "trace" / "**" / "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, dst / f"{tag}_kernel_stats.csv")
for line in open(src / "bench_trace.log"):
    if line.startswith("{"):
        (dst / f"{tag}_bench_under_rocprof.json").write_text(line)
rec = json.loads((src / "traffic_counters.json").read_text())
rec.update(workload=workload, gpus=gpus, path=path, source=f"profiles/{rnd}/{tag}_rocprofv3_summary.txt")
(root / "profiles" / "traffic.json").write_text(json.dumps(rec, indent=1) + "\n")
print(json.dumps(rec))
