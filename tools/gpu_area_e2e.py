#!/usr/bin/env python3
"""Where the end-to-end time of the product ld_area call goes beyond its scan (configs[2]): host clock around the pieces of
ops.ld_area for a kept plan -- graph replay (launch), the one host read, the result allocation + copy kernel, the final sync."""
import json
import statistics
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from ld_tools_amd import PackedPanel, ld_area, synth  # noqa: E402

n, h = 100000, 5008
p = PackedPanel.from_codes(synth.synth_codes_device(n, h))
pos = torch.as_tensor(synth.synth_positions(n, step=500)).to(p.device)
for _ in range(4):
    hits = ld_area(p, pos, None, 500000, "r_square", 0.8, check_positions=False)
torch.cuda.synchronize()
plan = [pl for k, pl in p._area_plans.items() if k != "all_rows"][0]
assert plan.graph
rows = {"replay_launch": [], "gpu_graph_ms": [], "host_read": [], "whole_call": []}
for _ in range(30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a.record()
    plan.graph.replay()
    b.record()
    t1 = time.perf_counter()
    tot = plan.summary.tolist()
    t2 = time.perf_counter()
    rows["replay_launch"].append((t1 - t0) * 1e3)
    rows["host_read"].append((t2 - t1) * 1e3)
    rows["gpu_graph_ms"].append(a.elapsed_time(b))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ld_area(p, pos, None, 500000, "r_square", 0.8, check_positions=False)
    torch.cuda.synchronize()
    rows["whole_call"].append((time.perf_counter() - t0) * 1e3)
print(json.dumps({k: round(statistics.median(v), 4) for k, v in rows.items()}))
