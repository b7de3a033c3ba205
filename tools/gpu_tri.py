#!/usr/bin/env python3
"""Time ld_triangle on a synthetic panel: python tools/gpu_tri.py <snps> <haps> <path> [reps] [ld32|k16]
(environment: SYNTH_MONO / SYNTH_MISS / SYNTH_MISS_ROWS = the generator's shares of monomorphic SNPs, of code-2 alleles and of
the rows that carry them: ld_tools_amd/synth.py)"""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from ld_tools_amd import PackedPanel, ld_triangle, ops, synth  # noqa: E402

n, h, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
fmt = sys.argv[5] if len(sys.argv) > 5 else "ld32"
ops.set_triangle_path(path)
p = PackedPanel.from_codes(synth.synth_codes_device(n, h, mono=float(os.environ.get("SYNTH_MONO", 0)),
                                                    miss=float(os.environ.get("SYNTH_MISS", 0)),
                                                    miss_rows=float(os.environ.get("SYNTH_MISS_ROWS", 1))))
res = ld_triangle(p, fmt=fmt)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    ld_triangle(p, out=res, fmt=fmt)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
out = {"snps": n, "haps": h, "path": path, "fmt": fmt, "ms": ms, "pairs_per_s": p.n_pairs / (ms * 1e-3)}
import ctypes  # noqa: E402
from ld_tools_amd._lib import lib  # noqa: E402
cnt = (ctypes.c_uint64 * 8)()
lib.ldx_debug_counters(cnt, 0)
if any(cnt):   # tuning builds only
    out["counters"] = {"f32_units": cnt[0], "parked_steps": cnt[1], "overflow_units": cnt[2], "mirror_evals": cnt[3],
                       "launches": reps + 1}
print(json.dumps(out), flush=True)
