#!/usr/bin/env python3
"""Sweep the number of half-height tail passes (ldx_debug_force_short_passes): python tools/gpu_short.py <snps> <haps> [reps]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from ld_tools_amd import PackedPanel, ld_triangle, synth  # noqa: E402
from ld_tools_amd._lib import lib  # noqa: E402

n, h = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
p = PackedPanel.from_codes(synth.synth_codes_device(n, h))
res = ld_triangle(p, fmt="k16", path="fp4")
for ns in (-1, 0, 32, 64, 128, 192, 256, 384, 512, 768, 1024):
    lib.ldx_debug_force_short_passes(ns)
    for _ in range(20):
        ld_triangle(p, out=res, fmt="k16", path="fp4")
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            ld_triangle(p, out=res, fmt="k16", path="fp4")
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    print(f"n_short {ns:5d}: {best:.4f} ms")
lib.ldx_debug_force_short_passes(-1)
