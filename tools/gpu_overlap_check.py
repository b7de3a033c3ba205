#!/usr/bin/env python3
"""Do consecutive ld_triangle launches of ONE stream ever overlap?  (round 5: a HIP graph whose kernel nodes alternate between two
result buffers returned wrong cells in half the time.)  python tools/gpu_overlap_check.py [snps] [haps] [batches]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from ld_tools_amd import PackedPanel, ld_triangle, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
h = int(sys.argv[2]) if len(sys.argv) > 2 else 5008
batches = int(sys.argv[3]) if len(sys.argv) > 3 else 40
p = PackedPanel.from_codes(synth.synth_codes_device(n, h))
ref = ld_triangle(p, fmt="k16")
outs = [ld_triangle(p, fmt="k16") for _ in range(2)]
torch.cuda.synchronize()


def run(nbuf, count):
    for k in range(count):
        ld_triangle(p, out=outs[k % nbuf], fmt="k16")


def timed(fn, label, nbuf):
    for o in outs:
        o.cells.view(torch.int16).fill_(-1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    ok = [bool(torch.equal(o.cells, ref.cells)) for o in outs[:nbuf]]
    print(f"{label}: {a.elapsed_time(b) / batches:.4f} ms per launch, results equal: {ok}", flush=True)


for nbuf in (1, 2):
    timed(lambda: run(nbuf, batches), f"eager, current stream, {nbuf} result buffer(s)", nbuf)
side = torch.cuda.Stream()
for nbuf in (1, 2):
    def on_side():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run(nbuf, batches)
        torch.cuda.current_stream().wait_stream(side)
    timed(on_side, f"eager, a side stream, {nbuf} result buffer(s)", nbuf)
for nbuf in (1, 2):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run(nbuf, batches)
    g.replay()
    torch.cuda.synchronize()
    timed(g.replay, f"one HIP graph, {nbuf} result buffer(s)", nbuf)
    timed(g.replay, f"one HIP graph, {nbuf} result buffer(s), again", nbuf)
