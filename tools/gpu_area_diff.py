#!/usr/bin/env python3
"""Where the band kernel's hits differ from the popcount scan's: python tools/gpu_area_diff.py [snps] [haps] [thres]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from ld_tools_amd import PackedPanel, ld_area, ops, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1008
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
p = PackedPanel.from_codes(synth.synth_codes_device(n, h))
pos = torch.as_tensor(synth.synth_positions(n, step=500)).to(p.device)
got = ld_area(p, pos, None, 20000, "r_square", thr, check_positions=False)
old = ops.get_area_path()
ops.set_area_path("popcount")
ref = ld_area(p, pos, None, 20000, "r_square", thr, check_positions=False)
ops.set_area_path(old)
g = set(zip(got.query.tolist(), got.oppos.tolist()))
r = set(zip(ref.query.tolist(), ref.oppos.tolist()))
import ctypes  # noqa: E402
from ld_tools_amd._lib import lib  # noqa: E402
cnt = (ctypes.c_uint64 * 8)()
lib.ldx_debug_counters(cnt, 0)
print("counters (tuning builds)", list(cnt))
print("hits", len(got), "ref", len(ref), "missing", len(r - g), "extra", len(g - r))
def show(tag, q, o):
    i, j = max(q, o), min(q, o)
    ri = i % 64
    print(f"{tag} query={q} oppos={o}  i={i} j={j} unit_row={i // 64} tile={j // 128} m={ri // 32} e={(ri % 32) % 4 + 4 * ((ri % 32) // 8)} half={((ri % 32) // 4) % 2} tt={(j % 128) // 32} l32={j % 32}")
for q, o in sorted(g - r)[:10]:
    show("extra", q, o)
for q, o in sorted(r - g)[:40]:
    show("missing", q, o)
    continue
    i, j = max(q, o), min(q, o)
    ri = i % 64
    print(f"missing query={q} oppos={o}  i={i} j={j} unit_row={i // 64} tile={j // 128} m={ri // 32} e={(ri % 32) % 4 + 4 * ((ri % 32) // 8)} half={((ri % 32) // 4) % 2} tt={(j % 128) // 32} l32={j % 32}")
