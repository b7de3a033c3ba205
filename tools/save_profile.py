#!/usr/bin/env python3
"""Copy the judged evidence of one tools/gpu_prof.sh run from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/save_profile.py <tag> <round-dir> <workload> [path] [gpus] [fmt] [--no-traffic]
    e.g.  r02a r02 "ld_triangle 10000x5008" fp4 1 k16      (writes profiles/traffic.json unless --no-traffic)
"""
import glob
import json
import shutil
import subprocess
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
argv = [a for a in sys.argv if a != "--no-traffic"]
tag, rnd, workload = argv[1], argv[2], argv[3]
path = argv[4] if len(argv) > 4 else "fp4"
gpus = int(argv[5]) if len(argv) > 5 else 1
fmt = argv[6] if len(argv) > 6 else "k16"
src = root / "gpurun_out" / f"prof_{tag}"
dst = root / "profiles" / rnd
dst.mkdir(parents=True, exist_ok=True)
shutil.copy(src / "summary.txt", dst / f"{tag}_rocprofv3_summary.txt")
stats = sorted(glob.glob(str(src / "trace" / "**" / "*kernel_stats.csv"), recursive=True), key=lambda f: Path(f).stat().st_mtime)
if stats:   # gpurun_out/ keeps the files of earlier runs with the same tag: the newest one is this run's
    shutil.copy(stats[-1], dst / f"{tag}_kernel_stats.csv")
for line in open(src / "bench_trace.log"):
    if line.startswith("{"):
        (dst / f"{tag}_bench_under_rocprof.json").write_text(line)
if (src / "command.txt").exists():
    shutil.copy(src / "command.txt", dst / f"{tag}_command.txt")
if (src / "launch_durations.json").exists():    # per-launch mean / median / min and the steady back-to-back run (prof_summary.py)
    shutil.copy(src / "launch_durations.json", dst / f"{tag}_launch_durations.json")
# round 5: where the cycles go (tools/pmc_table.py over the sq / sq2 / sq3 / sq4 counter passes)
tab = subprocess.run([sys.executable, str(root / "tools" / "pmc_table.py"), str(src), "triangle_mfma_kernel"], capture_output=True, text=True)
if tab.returncode == 0 and tab.stdout.strip():
    (dst / f"{tag}_cycle_accounting.txt").write_text(tab.stdout)
    if (src / "pmc_table.json").exists():
        shutil.copy(src / "pmc_table.json", dst / f"{tag}_counters.json")
else:
    print("no cycle table:", tab.stderr.strip()[-300:])
if (src / "traffic_counters.json").exists():
    rec = json.loads((src / "traffic_counters.json").read_text())
    sys.path.insert(0, str(root))
    from ld_tools_amd.build import source_digest
    try:    # the commit whose tree went to the GPU box (run this right after the gpurun call, before editing on)
        head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "ld_tools_amd/csrc", "include"], cwd=root,
                                    capture_output=True, text=True).stdout.strip())
    except OSError:
        head, dirty = None, None
    rec.update(workload=workload, gpus=gpus, path=path, fmt=fmt, profile=tag,
               source=f"profiles/{rnd}/{tag}_rocprofv3_summary.txt", head=head, kernel_sources_dirty=dirty,
               kernel_src=source_digest())
    (dst / f"{tag}_traffic.json").write_text(json.dumps(rec, indent=1) + "\n")
    if "--no-traffic" not in sys.argv:
        (root / "profiles" / "traffic.json").write_text(json.dumps(rec, indent=1) + "\n")
    print(json.dumps(rec))
