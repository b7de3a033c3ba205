#!/usr/bin/env python3
"""Which workgroups are the fast ones?  passes per workgroup against blockIdx half and against launch order on the CU."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
grid, waves, stride, passes = (int(x) for x in raw[:4])
st = raw[4:].reshape(grid * waves, stride)
w0 = st[::waves]                     # wave 0 of every workgroup
npass = w0[:, 3].astype(int)
hw = w0[:, 0]
cu = ((hw >> 32) & 0xF) << 8 | ((hw >> 13) & 7) << 5 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 0xF)
half = (np.arange(grid) >= grid // 2).astype(int)
for h in (0, 1):
    sel = half == h
    print(f"blockIdx half {h}: workgroups {sel.sum()}  passes mean {npass[sel].mean():.2f}  min {npass[sel].min()}  max {npass[sel].max()}")
# per CU: the workgroup with more passes -- is it the one with the lower blockIdx?
lower_wins = ties = n = 0
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    if len(idx) != 2:
        continue
    n += 1
    a, b = idx            # a < b
    if npass[a] > npass[b]: lower_wins += 1
    elif npass[a] == npass[b]: ties += 1
print(f"CUs with two workgroups: {n}; lower blockIdx did more passes on {lower_wins}, ties {ties}")
t_end = w0[:, 4].astype(np.int64)
t_end = (t_end - w0[:, 1].astype(np.int64).min()) / 100.0
for h in (0, 1):
    sel = half == h
    print(f"half {h}: end time us p50 {np.percentile(t_end[sel], 50):.1f} max {t_end[sel].max():.1f}")
