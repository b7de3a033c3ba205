#!/usr/bin/env python3
"""After `bash tools/gpu_prof_all.sh <R>` on the GPU box: copy the judged files of the four profiles and the default bench
line into profiles/<R>/ and refresh profiles/traffic.json (with the commit and the kernel-source digest it was taken at).

    python tools/save_profile_all.py r03
"""
import json
import shutil
import subprocess
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
R = sys.argv[1] if len(sys.argv) > 1 else "r03"
py = sys.executable
runs = [(f"{R}a", "ld_triangle 10000x5008", []), (f"{R}b", "ld_triangle 40000x5008", ["--no-traffic"]),
        (f"{R}c", "ld_triangle 50000x1008", ["--no-traffic"]), (f"{R}d", "ld_area 100000 +-500kb r2>=0.8", ["--no-traffic"])]
for tag, workload, extra in runs:
    if not (root / "gpurun_out" / f"prof_{tag}" / "summary.txt").exists():
        print(f"{tag}: no summary, skipped")
        continue
    subprocess.run([py, str(root / "tools" / "save_profile.py"), tag, R, workload, "fp4", "1", "k16", *extra], check=True)
src = root / "gpurun_out" / f"bench_default_{R}.json"
if src.exists() and src.stat().st_size:
    # The default bench ran on the GPU box BEFORE this script refreshed profiles/traffic.json, so its line carries the
    # previous record's traffic / digest.  Re-stamp the provenance from the record just written -- the
    # counters of THIS visit, same tree -- and say so in the line.
    text = src.read_text()
    dst = root / "profiles" / R / f"{R}a_bench_default.json"
    try:
        line = next(ln for ln in reversed(text.splitlines()) if ln.startswith("{"))
        rec = json.loads(line)
        tr = json.loads((root / "profiles" / "traffic.json").read_text())
        if tr.get("workload") == rec["config"]["workload"] and tr.get("profile") == f"{R}a":
            for obj in (rec.get("roofline", {}), rec.get("roofline_hbm", {})):
                obj.update(traffic=tr["hbm_bytes_per_launch"], traffic_head=tr.get("head"),
                           traffic_kernel_src=tr.get("kernel_src"),
                           traffic_is_of_this_kernel=tr.get("kernel_src") == obj.get("kernel_src"),
                           traffic_source=f"profiles/traffic.json ({R}a; rocprofv3 PMC of the same visit; re-stamped by "
                                          "tools/save_profile_all.py after the run)")
        dst.write_text(json.dumps(rec) + "\n")
    except (StopIteration, ValueError, KeyError, OSError) as exc:
        print("could not re-stamp the default bench line:", exc)
        shutil.copy(src, dst)
    print("bench default line ->", f"profiles/{R}/{R}a_bench_default.json")
