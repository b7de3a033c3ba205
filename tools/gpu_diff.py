#!/usr/bin/env python3
"""Where does one triangle path differ from the popcount kernel?  python tools/gpu_diff.py <snps> <haps> [fmt] [path]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from ld_tools_amd import PackedPanel, ld_triangle, synth  # noqa: E402

n, h = int(sys.argv[1]), int(sys.argv[2])
fmt = sys.argv[3] if len(sys.argv) > 3 else "k16"
path = sys.argv[4] if len(sys.argv) > 4 else "fp4"
p = PackedPanel.from_codes(synth.synth_codes_device(n, h, seed=synth.BENCH_SEED))
want = ld_triangle(p, fmt=fmt, path="popcount").cells
got = ld_triangle(p, fmt=fmt, path=path).cells
w = want.view(torch.int16 if fmt == "k16" else torch.int32).view(-1, 8192, 2)
g = got.view(torch.int16 if fmt == "k16" else torch.int32).view(-1, 8192, 2)
neq = (w != g).any(dim=2)
print("cells differing:", int(neq.sum()), "of", neq.numel(), "in units", int(neq.any(dim=1).sum()))
idx = neq.nonzero()[:24].cpu().tolist()
for u, c in idx:
    print(" unit64", u, "row", c // 128, "col", c % 128, "want", want.view(-1, 8192, 2)[u, c].tolist(), "got", got.view(-1, 8192, 2)[u, c].tolist())
rows = torch.unique(neq.nonzero()[:, 1] // 128).cpu().tolist()
print("rows hit:", rows[:64])
