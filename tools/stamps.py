#!/usr/bin/env python3
"""Summarise the in-kernel stamps of a tuning build (LDX_STAMPS=file): placement, clock, cycles per phase."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
grid, waves, stride, passes = (int(x) for x in raw[:4])
st = raw[4:].reshape(grid * waves, stride)
st = st[st[:, 3] > 0]
hw = st[:, 0]
wave_id = hw & 0xF; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = (hw >> 32) & 0xF
cu_key = (xcc << 8) | (se << 5) | (sh << 4) | cu
simd_key = (cu_key << 2) | simd
print(f"waves with work: {len(st)}  distinct CUs: {len(np.unique(cu_key))}  distinct SIMDs: {len(np.unique(simd_key))}")
_, cnt = np.unique(simd_key, return_counts=True)
print("waves per SIMD histogram:", dict(zip(*np.unique(cnt, return_counts=True))))
clk = (st[:, 5] - st[:, 2]).astype(np.float64) / ((st[:, 4] - st[:, 1]).astype(np.float64) / 100e6) / 1e9
print(f"in-kernel clock GHz: median {np.median(clk):.3f}  min {clk.min():.3f}  max {clk.max():.3f}")
life = (st[:, 5] - st[:, 2]).astype(np.float64)
print(f"wave lifetime cycles: median {np.median(life):.0f}  max {life.max():.0f}")
np_ = np.minimum(st[:, 3].astype(int), passes)
k, e, g, pr = [], [], [], []
for row, n in zip(st, np_):
    s = row[6:6 + 4 * n].reshape(n, 4).astype(np.int64)
    pr.append(s[:, 1] - s[:, 0]); k.append(s[:, 2] - s[:, 1]); e.append(s[:, 3] - s[:, 2])
    if n > 1:
        g.append(s[1:, 0] - s[:-1, 3])
k, e, pr = np.concatenate(k), np.concatenate(e), np.concatenate(pr)
q = lambda a: "  ".join(f"p{p}={np.percentile(a, p):.0f}" for p in (5, 25, 50, 75, 95))
print("prologue cycles per pass: ", q(pr), f" mean={pr.mean():.0f}")
print("K loop cycles per pass:   ", q(k), f" mean={k.mean():.0f}")
print("epilogue cycles per pass: ", q(e), f" mean={e.mean():.0f}")
if g:
    g = np.concatenate(g); print("gap between passes:       ", q(g))
print("passes per wave histogram:", dict(zip(*np.unique(st[:, 3].astype(int), return_counts=True))))
t0 = st[:, 2].astype(np.int64); tmin = t0.min()
print(f"wave start spread: p50={np.percentile(t0 - tmin, 50):.0f} max={(t0 - tmin).max():.0f} cycles;  kernel span (first start -> last end): {(st[:, 5].astype(np.int64).max() - tmin)} cycles")
# per-pass totals by pass index
for idx in range(int(st[:, 3].max())):
    sel = st[st[:, 3] > idx]
    s4 = sel[:, 6 + 4 * idx: 10 + 4 * idx].astype(np.int64)
    print(f"  pass #{idx}: n={len(sel):5d} start(p50, rel)={np.median(s4[:, 0] - tmin):9.0f}  K={np.median(s4[:, 2] - s4[:, 1]):7.0f}  E={np.median(s4[:, 3] - s4[:, 2]):7.0f}  end(p50)={np.median(s4[:, 3] - tmin):9.0f} end(max)={(s4[:, 3] - tmin).max():9.0f}")
    if idx >= 5: break

# per-step stamps of the fp32 tier (builds with -DLDX_TUNING -DLDX_STAMPS_ONLY): the second pass of every wave
if stride >= 6 + 4 * passes + 17:
    ss = st[:, 6 + 4 * passes: 6 + 4 * passes + 17].astype(np.int64)
    ss = ss[(ss[:, 0] > 0) & (ss[:, 16] > ss[:, 0])]
    if len(ss):
        d = np.diff(ss, axis=1)
        print(f"fp32-tier steps of the second pass ({len(ss)} waves): cycles per step  " + q(d.ravel()) + f"  mean={d.mean():.0f}")
        print("   by step index (median): " + " ".join(f"{np.median(d[:, k]):.0f}" for k in range(16)))
        print("   sixteen steps together: " + q(ss[:, 16] - ss[:, 0]))

# in-chunk stamps (builds with -DLDX_CHUNK_STAMPS): six s_memtime points of one chunk of each wave's second pass
cs = st[:, 6 + 4 * (passes - 2): 6 + 4 * (passes - 2) + 6].astype(np.int64)
cs = cs[(cs[:, 0] > 0) & (cs[:, 5] > cs[:, 0])]
if len(cs) and st[:, 3].max() < passes - 2:
    d = np.diff(cs, axis=1)
    names = ["loads+step0", "step1", "step2", "barrier wait", "step3"]
    print("in-chunk segments (cycles, median / p95) over", len(cs), "waves:")
    for k, nm in enumerate(names):
        print(f"   {nm:14s} {np.median(d[:, k]):7.0f} {np.percentile(d[:, k], 95):7.0f}")
    print(f"   {'chunk total':14s} {np.median(cs[:, 5] - cs[:, 0]):7.0f}")

# end-of-kernel balance from the constant-rate counter (s_memrealtime, 100 MHz, chip-wide)
rt0, rt1 = st[:, 1].astype(np.int64), st[:, 4].astype(np.int64)
span = (rt1.max() - rt0.min()) / 100.0
ends = (rt1 - rt0.min()) / 100.0
print(f"kernel span {span:.1f} us; wave end times (us): p5={np.percentile(ends,5):.1f} p25={np.percentile(ends,25):.1f} "
      f"p50={np.percentile(ends,50):.1f} p75={np.percentile(ends,75):.1f} p95={np.percentile(ends,95):.1f} max={ends.max():.1f}")
print(f"mean busy fraction of a wave slot: {np.mean((rt1 - rt0) / 100.0) / span:.3f}")
for npass_ in sorted(set(st[:, 3].astype(int))):
    sel = st[:, 3].astype(int) == npass_
    print(f"   waves with {npass_} passes: {sel.sum():5d}  end time median {np.median(ends[sel]):.1f} us  max {ends[sel].max():.1f} us")
