#!/usr/bin/env python3
"""One stream against two, under the SAME conditions: python tools/gpu_streams.py <snps> <haps> [batches] [rounds]

Both modes run `batches` independent ld_triangle launches as ONE HIP graph (mode 1: a chain on the capturing stream, what
bench.py's headline does; mode 2: alternating on two side streams with a fork / join around them, what
bench.py's other_paths.two_streams does), after the same settling load, interleaved `rounds` times on the same box.
Printed: ms per batch of each mode and round, and their medians.  Results of both modes are compared with a reference.
"""
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from ld_tools_amd import PackedPanel, ld_triangle, synth  # noqa: E402

n, h = int(sys.argv[1]), int(sys.argv[2])
batches = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
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


graphs = {}
for name, fn in (("one", one), ("two", two)):
    fn(4)                      # every stream has launched before its capture (ticket-counter slots)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn(batches)
    g.replay()
    torch.cuda.synchronize()
    graphs[name] = g

for _ in range(5):             # settle: ~0.15 s of load
    graphs["one"].replay()
torch.cuda.synchronize()
res = {"one": [], "two": []}
for r in range(rounds):
    for name in ("one", "two"):
        for o in outs:
            o.cells.view(torch.int16).fill_(-1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graphs[name].replay()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / batches
        ok = all(torch.equal(o.cells, ref.cells) for o in outs)
        res[name].append(ms)
        print(f"round {r} {name:3s} stream(s): {ms:.4f} ms per batch, results equal: {ok}", flush=True)
m1, m2 = statistics.median(res["one"]), statistics.median(res["two"])
print(f"{n}x{h}, {batches} batches per graph: median one stream {m1:.4f} ms, two streams {m2:.4f} ms ({(m2 / m1 - 1) * 100:+.1f} %)")
