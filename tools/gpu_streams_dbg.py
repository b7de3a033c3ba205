#!/usr/bin/env python3
"""The sequence that showed round 5's bug: V=0 python tools/gpu_streams_dbg.py

A one-stream HIP graph of ld_triangle launches into ALTERNATING result buffers, captured BEFORE two side streams are used
eagerly and a fork / join graph is captured (V=0, V=3); then replayed.  With the ticket counters re-armed by plain stores
every launch of the one-stream graph after its first drew no ticket beyond its static one (results equal: [True, False]
at half the time).  V=1 captures the graphs in the other order, V=2 only the one-stream graph: both were always right.
Now a GPU test (test_graph_of_launches_into_alternating_buffers)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from ld_tools_amd import PackedPanel, ld_triangle, synth  # noqa: E402

V = os.environ.get("V", "0")
n, h, batches = 10000, 5008, 6
p = PackedPanel.from_codes(synth.synth_codes_device(n, h))
ref = ld_triangle(p, fmt="k16")
outs = [ld_triangle(p, fmt="k16"), ld_triangle(p, fmt="k16")]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()


def one(count):
    for k in range(count):
        ld_triangle(p, out=outs[k & 1], fmt="k16")


def two(count):
    cur = torch.cuda.current_stream()
    for st in streams:
        st.wait_stream(cur)
    for k in range(count):
        with torch.cuda.stream(streams[k & 1]):
            ld_triangle(p, out=outs[k & 1], fmt="k16")
    for st in streams:
        cur.wait_stream(st)


def check(g, label):
    for o in outs:
        o.cells.view(torch.int16).fill_(-1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    ok = [bool(torch.equal(o.cells, ref.cells)) for o in outs]
    print(f"V={V}: {label}: {a.elapsed_time(b) / batches:.4f} ms per batch, results equal: {ok}", flush=True)


order = (("two", two), ("one", one)) if V == "1" else ((("one", one),) if V == "2" else (("one", one), ("two", two)))
graphs = {}
for name, fn in order:
    fn(4)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    g.enable_debug_mode()
    with torch.cuda.graph(g):
        fn(batches)
    os.makedirs("gpurun_out/dot", exist_ok=True)
    g.debug_dump(f"gpurun_out/dot/V{V}_{name}.dot")
    if V == "3":
        check(g, f"{name} right after its capture")
    else:
        g.replay()
        torch.cuda.synchronize()
    graphs[name] = g
for r in range(2):
    for name in graphs:
        check(graphs[name], f"round {r} {name}")
