#!/usr/bin/env python3
"""Independent ld_triangle batches on S streams (each with its own result buffer): python tools/gpu_pipe.py <snps> <haps> [reps]
The tail of one launch (workgroups draining their last passes) overlaps the start of the next batch's launch."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from ld_tools_amd import PackedPanel, ld_triangle, synth  # noqa: E402

n, h = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
p = PackedPanel.from_codes(synth.synth_codes_device(n, h))
ref = ld_triangle(p, fmt="k16")
for S in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [ld_triangle(p, fmt="k16") for _ in range(S)]
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        for o in outs:
            o.cells.view(torch.int16).fill_(-1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        for k in range(reps):
            with torch.cuda.stream(streams[k % S]):
                ld_triangle(p, out=outs[k % S], fmt="k16")
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    ok = all(torch.equal(o.cells, ref.cells) for o in outs)
    print(f"{S} stream(s): {best:.4f} ms per batch, {p.n_pairs / best / 1e-3:.3e} pairs/s, results equal: {ok}")
