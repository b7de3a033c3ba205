#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(root, pattern), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield f, r


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
seen = False
for f, r in rows("trace/**/*kernel_stats.csv"):
    seen = True
    print(f"{r.get('Name', '')[:70]:70s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} "
          f"avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")

# Per-launch durations from the kernel trace: the --stats average mixes cold / eager warm-up launches with the steady
# state (its mean exceeded the bench's own ms_per_step).  Beside mean / median / min of ALL launches of a
# kernel, "steady" = the launches of its longest back-to-back run (next start within 50 us of the previous end: the
# graph replays of bench.py's settle + timed region, or a long eager loop), first tenth dropped (clock settling).
launches = defaultdict(list)
for f, r in rows("trace/**/*kernel_trace.csv"):
    launches[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


steady_json = {}
if launches:
    print("== per-launch durations from the kernel trace (ns) ==")
    for name, iv in sorted(launches.items(), key=lambda kv: -sum(e - b for b, e in kv[1])):
        iv.sort()
        d = [e - b for b, e in iv]
        line = (f"{name[:70]:70s} n={len(d)} mean={sum(d) / len(d):.0f} median={med(d):.0f} min={min(d)} max={max(d)}")
        best, cur = [], [0]
        for k in range(1, len(iv)):
            if iv[k][0] - iv[k - 1][1] <= 50000:
                cur.append(k)
            else:
                best, cur = (cur if len(cur) > len(best) else best), [k]
        best = cur if len(cur) > len(best) else best
        if len(best) >= 10:
            run = [d[k] for k in best[len(best) // 10:]]
            line += f" | steady run of {len(best)}: mean={sum(run) / len(run):.0f} median={med(run):.0f} min={min(run)}"
            steady_json[name] = {"launches": len(d), "mean_ns": sum(d) / len(d), "median_ns": med(d), "min_ns": min(d),
                                 "steady_run": len(best), "steady_mean_ns": sum(run) / len(run), "steady_median_ns": med(run)}
        if not seen or "triangle" in name or "area" in name or "pack" in name:
            print(line)

for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_sq4"):
    acc = defaultdict(lambda: defaultdict(list))
    for f, r in rows(f"{tag}/**/*counter_collection.csv"):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        print(f"== {tag} (per dispatch mean) ==")
        for k, cs in acc.items():
            if "triangle" not in k and "pack" not in k and "area" not in k:
                continue
            print(k[:90])
            for c, v in cs.items():
                print(f"    {c:24s} n={len(v):4d} mean={sum(v) / len(v):.4g}")


# ---- HBM traffic of the dominant kernel -> profiles-ready JSON (bench.py reads profiles/traffic.json) ----
# MI355X_MICROARCH.md, HBM: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts a
# wide coalesced read at half its bytes (x2); WRITE_SIZE is exact for our stores (r01a: equals the output size).
import json

def mean_counter(tag, counter, needle):
    """Mean over the dispatches of the MOST-LAUNCHED kernel whose name contains `needle` (the timed instantiation: the
    other cell format and the int8 comparison legs are other instantiations of the same template and write other sizes)."""
    by = defaultdict(list)
    for f, r in rows(f"{tag}/**/*counter_collection.csv"):
        if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
            by[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    if not by:
        return None
    v = max(by.values(), key=len)
    return sum(v) / len(v)

for needle in ("triangle_mfma_kernel", "triangle_kernel"):
    fetch, write = mean_counter("pmc_fetch", "FETCH_SIZE", needle), mean_counter("pmc_write", "WRITE_SIZE", needle)
    if fetch is not None and write is not None:
        rec = {"kernel": needle, "fetch_size_kib": fetch, "write_size_kib": write,
               "hbm_bytes_per_launch": 2.0 * fetch * 1024.0 + write * 1024.0,
               "correction": "FETCH_SIZE x2 (gfx950 wide-read tally), WRITE_SIZE as reported"}
        print("== traffic ==")
        print(json.dumps(rec))
        with open(os.path.join(root, "traffic_counters.json"), "w") as fh:
            json.dump(rec, fh)
        break
if steady_json:
    with open(os.path.join(root, "launch_durations.json"), "w") as fh:
        json.dump(steady_json, fh, indent=1)
