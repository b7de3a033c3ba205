#!/usr/bin/env python3
"""Round 6, VERDICT r05 item 5: what the "k and the fraction from ONE add" margin test would park, simulated on the CPU.

    python tools/margin_sim.py [snps] [haps]

For all row > col pairs of the first `snps` SNPs of the bench panel (ld_tools_amd/synth.py, host generator): y_r = 10^4 r^2 and
y_d = 10^4 D' in float32 as the fp32 tier computes them, then two "sure" tests per value:
  * shipped:   |y - rint(y)| + eta y < 1/2 - c0                      (csrc/ldx_common.h, ld_multi_f32)
  * candidate: z = y + (2^14 + 1/2) in float32 leaves nine fraction bits f9; sure iff min(f9, 511 - f9) >= t with
               t = ceil((eta y + c0 + 2^-10) 2^9)                    (the add's own rounding is the 2^-10)
and the share of LANE-STEPS (8 pairs = 16 values: two rows x four columns 32 apart, as a lane holds them) each would park.
No GPU, no library: numpy only."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("synth", Path(__file__).resolve().parent.parent / "ld_tools_amd" / "synth.py")
synth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synth)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
h = int(sys.argv[2]) if len(sys.argv) > 2 else 5008
codes = synth.synth_codes_host(n, h)
g = (codes == 1).astype(np.float32)
a = g.sum(axis=1).astype(np.float64)
r = h - a
ok = (a > 0) & (r > 0)
c = (g @ g.T).astype(np.float64)                       # n11, exact (< 2^24)
f32 = np.float32
dn = (h * c - np.outer(a, a)).astype(f32)              # exact integers
with np.errstate(divide="ignore", invalid="ignore"):
    s = (10.0 / np.sqrt(a * r)).astype(f32)
    ra_s, rr_s = (1e4 / a).astype(f32), (1e4 / r).astype(f32)
    ra, rr = (1.0 / a).astype(f32), (1.0 / r).astype(f32)
    t = (dn * s[:, None]).astype(f32) * s[None, :]
    yr = (t * t).astype(f32)
    neg = dn < 0
    x = np.where(neg, ra[None, :], rr[None, :]) * ra_s[:, None]
    y = np.where(neg, rr[None, :], ra[None, :]) * rr_s[:, None]
    yd = (np.abs(dn) * np.maximum(x, y).astype(f32)).astype(f32)
u = 2.0 ** -24
eta_r, eta_d = 14 * u, 7 * u
c0 = 6e-12 * h * h + 4e-11 * h * h / (h - 1) + 2e-6


def shipped(yv, eta):
    return np.abs(yv - np.rint(yv)) + eta * yv < 0.5 - c0


def candidate(yv, eta):
    z = (yv.astype(f32) + f32(2.0 ** 14 + 0.5)).astype(f32)          # one float32 add
    f9 = np.floor((z.astype(np.float64) % 1.0) * 512.0)
    tt = np.ceil((eta * yv + c0 + 2.0 ** -10) * 512.0)
    return np.minimum(f9, 511.0 - f9) >= tt


valid = np.tril(np.ones((n, n), dtype=bool), -1) & ok[:, None] & ok[None, :] & (dn != 0)
res = {}
for name, fn in (("shipped", shipped), ("candidate", candidate)):
    sure = fn(yr.astype(np.float64), eta_r) & fn(yd.astype(np.float64), eta_d)
    sure = sure | ~valid                                           # pairs outside the triangle / Dn == 0 are not this test's business
    # lane-steps: rows (i, i + 32 within a 64-row unit is the kernel's pairing; any fixed pairing has the same statistics) x
    # columns (j, j + 32, j + 64, j + 96) of a 128-column tile
    nn = n // 128 * 128
    blk = sure[:nn, :nn].reshape(nn // 2, 2, nn // 128, 4, 32)       # (row pair, 2, tile, tt, l32)
    step_sure = blk.all(axis=(1, 3))
    inside = np.tril(np.ones((nn // 2, nn // 128), dtype=bool), -1)[:, :, None] & np.ones(32, dtype=bool)   # rough: steps below the diagonal
    rows_lo = (np.arange(nn // 2) * 2)[:, None, None]
    cols_hi = (np.arange(nn // 128) * 128 + 127)[None, :, None]
    inside = np.broadcast_to(rows_lo > cols_hi, step_sure.shape)
    v = valid[:nn, :nn]
    res[name] = (1.0 - sure[:nn, :nn][v].mean(), 1.0 - step_sure[inside].mean())
    print(f"{name:10s}: values not sure (either measure) {100 * res[name][0]:.3f} % of the pairs, lane-steps parked {100 * res[name][1]:.2f} %")
print(f"candidate / shipped lane-steps: x{res['candidate'][1] / max(res['shipped'][1], 1e-12):.1f}   ({n} SNPs x {h} haplotypes)")
