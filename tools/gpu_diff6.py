#!/usr/bin/env python3
"""Round 6 debugging aid: where the FP4 product variant differs from the popcount kernel, per cell, with the SNPs' classes.
python tools/gpu_diff6.py <snps> <haps> [fmt] [mono] [miss] [miss_rows]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ld_tools_amd import PackedPanel, ld_triangle, synth  # noqa: E402
from ld_tools_amd._lib import cell_offset  # noqa: E402

n, h = int(sys.argv[1]), int(sys.argv[2])
fmt = sys.argv[3] if len(sys.argv) > 3 else "k16"
mono = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
miss = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
mrows = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0
p = PackedPanel.from_codes(synth.synth_codes_device(n, h, seed=synth.BENCH_SEED, mono=mono, miss=miss, miss_rows=mrows))
a = p.acnt[:n].cpu().numpy().view(np.uint32).astype(np.int64)
r = p.rcnt[:n].cpu().numpy().view(np.uint32).astype(np.int64)
cls = np.where((a == 0) | (r == 0), 1, np.where(8 * (h - a - r) <= r, 0, 2))   # csrc/ldx_common.h, snp_class
print("classes: ordinary", int((cls == 0).sum()), "degenerate", int((cls == 1).sum()), "odd", int((cls == 2).sum()))
want = ld_triangle(p, fmt=fmt, path="popcount")
got = ld_triangle(p, fmt=fmt, path="fp4")
torch.cuda.synchronize()
view = torch.int16 if fmt == "k16" else torch.int32
w = want.cells.view(view).cpu().numpy().reshape(-1, 2)
g = got.cells.view(view).cpu().numpy().reshape(-1, 2)
neq = (w != g).any(axis=1)
print("cells differing:", int(neq.sum()), "of", neq.size)
if neq.any():
    # invert the cell index: for every (row > col) pair its element
    rows, cols = np.tril_indices(n, -1)
    idx = got.cell_index(rows, cols)
    bad = neq[idx]
    print("valid cells differing:", int(bad.sum()), " (the rest are padding / upper-triangle cells)")
    br, bc = rows[bad], cols[bad]
    print("by class (row, col):", {(int(x), int(y)): int(((cls[br] == x) & (cls[bc] == y)).sum()) for x in range(3) for y in range(3)})
    for k in range(min(30, br.size)):
        i = idx[bad][k]
        print(f" row {br[k]} (cls {cls[br[k]]}, a {a[br[k]]}) col {bc[k]} (cls {cls[bc[k]]}, a {a[bc[k]]}) unit {i // 1024} want {w[i].tolist()} got {g[i].tolist()}")
    units = np.unique(idx[bad] // 1024)
    print("units hit:", units.size, units[:40].tolist())
    rr = np.unique(br)
    cc = np.unique(bc)
    print("rows hit:", rr.size, rr[:40].tolist())
    print("cols hit:", cc.size, cc[:40].tolist())
    print("row % 64 histogram:", np.bincount(br % 64, minlength=64).tolist())
    print("col % 128 // 32 histogram:", np.bincount((bc % 128) // 32, minlength=4).tolist())
    # the first affected (64-row block, tile): which steps / lanes, and where its degenerate SNPs sit
    b0, t0 = int(br[0]) // 64, int(bc[0]) // 128
    sel = (br // 64 == b0) & (bc // 128 == t0)
    ri, cj = br[sel] % 64, bc[sel] % 128
    rem = ri % 32
    steps = sorted(set(zip(((rem & 3) + 4 * (rem >> 3)).tolist(), (ri // 32).tolist(), ((rem >> 2) & 1).tolist(), (cj // 32).tolist(), (cj % 32).tolist())))
    print(f"block rows {b0 * 64}..{b0 * 64 + 63} x tile {t0}: {len(steps)} wrong cells as (e, m, half, tt, l32):", steps[:80])
    dr = [int(x) for x in np.nonzero(cls[b0 * 64:b0 * 64 + 64] == 1)[0]]
    dc = [int(x) for x in np.nonzero(cls[t0 * 128:t0 * 128 + 128] == 1)[0]]
    print("degenerate rows of the block (ri):", dr, " as (e, m, half):", [((x % 32 & 3) + 4 * (x % 32 >> 3), x // 32, (x % 32 >> 2) & 1) for x in dr])
    print("degenerate cols of the tile (c):", dc, " as (tt, l32):", [(x // 32, x % 32) for x in dc])
    tiles = np.unique(bc // 128)
    print("tiles hit:", tiles.tolist(), " tiles with a degenerate column:", np.unique(np.nonzero(cls == 1)[0] // 128).tolist())
