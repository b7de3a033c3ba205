#!/usr/bin/env python3
"""Randomised cross-check of the matrix-pipe kernels against the AND+popcount kernels, for a bounded time.

    python tools/gpu_fuzz.py [seconds] [seed]

Every round draws a panel shape (SNPs 2..9000, haplotypes 16..6000, a missing-code rate, sometimes monomorphic or
all-missing rows), packs it, and compares
  * ld_triangle: 'fp4' and 'mfma' against 'popcount', both cell formats, with and without the n11 plane, on the whole
    triangle and on a random unit range, each matrix-pipe launch repeated (the second launch into a poisoned buffer); the
    one-measure cells (round 6) against the halves of the 4-byte cells; one panel in five with 5 % or 30 % monomorphic SNPs;
  * ld_pairs, pair_counts and the fused drop-in calc_ld on random pairs against the triangle's cells and n11 plane;
  * ld_area: the three kernels' ordered hit lists for a random flank / measure / threshold / query subset;
  * (one round in twenty) a HIP graph of two to five matrix-kernel launches into two alternating result buffers, some of
    them forked onto a side stream, replayed twice.
Any difference is printed with its shape and seed and the process exits 1.  The popcount kernels are the independent
second implementation (they share no counting or staging code with the matrix-pipe kernel) and are themselves pinned to
the oracle and the reference's golden outputs by tests/test_gpu_parity.py.
"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ld_tools_amd import PackedPanel, ld_area, ld_triangle, ops, pair_counts, synth  # noqa: E402
from ld_tools_amd.ops import ld_pairs  # noqa: E402
from ld_tools_amd.backend.calc_ld import calc_ld  # noqa: E402


class Mismatch(AssertionError):
    pass


def run(budget: float = 120.0, seed: int = 1, progress: float = 0.0) -> str:
    rng = np.random.RandomState(seed)
    t_end = time.time() + budget
    t_say = time.time() + progress
    rounds = pairs = hits = graphs = 0

    def fail(msg):
        raise Mismatch(msg)

    while time.time() < t_end:
        n = int(rng.choice([rng.randint(2, 130), rng.randint(130, 700), rng.randint(700, 3000), rng.randint(3000, 9000)]))
        h = int(rng.choice([rng.randint(16, 300), rng.randint(300, 1100), rng.choice([1008, 2504, 5008, 5096]), rng.randint(1100, 6000)]))
        miss = float(rng.choice([0.0, 0.0, 0.002, 0.05]))
        seed = int(rng.randint(1, 1 << 30))
        # (round 6) sometimes a panel that is far from "all ordinary": a share of monomorphic SNPs, missing codes in a share of rows
        mono = float(rng.choice([0.0, 0.0, 0.0, 0.05, 0.3]))
        mrows = float(rng.choice([1.0, 1.0, 0.2]))
        codes = synth.synth_codes_device(n, h, seed=seed, miss=miss, mono=mono, miss_rows=mrows)
        if rng.rand() < 0.3:                                   # degenerate rows: monomorphic ALT / REF, all missing
            for r in rng.randint(0, n, size=3):
                codes[int(r), :h] = int(rng.choice([0, 1, 2]))
        p = PackedPanel.from_codes(codes)
        tag = f"n={n} h={h} miss={miss} mono={mono} miss_rows={mrows} seed={seed}"
        for fmt in ("k16", "ld32"):
            view = torch.int16 if fmt == "k16" else torch.int32
            want_n11 = bool(rng.rand() < 0.3)
            total = p.n_units
            ur = None
            if rng.rand() < 0.4:
                a = int(rng.randint(0, total))
                ur = (a, int(rng.randint(a, total + 1)))
            ref = ld_triangle(p, fmt=fmt, path="popcount", want_n11=want_n11, unit_range=ur)
            for path in ("fp4", "mfma"):
                got = ld_triangle(p, fmt=fmt, path=path, want_n11=want_n11, unit_range=ur)
                for rep in range(2):
                    if not torch.equal(got.cells.view(view), ref.cells.view(view)):
                        fail(f"ld_triangle {path} {fmt} n11={want_n11} unit_range={ur} launch {rep}: {tag}")
                    if want_n11 and not torch.equal(got.n11, ref.n11):
                        fail(f"ld_triangle n11 plane {path} {fmt} unit_range={ur}: {tag}")
                    got.cells.view(view).fill_(-1)
                    ld_triangle(p, fmt=fmt, path=path, want_n11=want_n11, unit_range=ur, out=got)
            pairs += ref.cells.shape[0]
            if fmt == "k16" and ur is None and rng.rand() < 0.5:   # (round 6) the one-measure cells against the halves of the 4-byte cells
                for col, f1 in ((0, "k16r"), (1, "k16d")):
                    for path in ("fp4", "popcount"):
                        one = ld_triangle(p, fmt=f1, path=path)
                        if not torch.equal(one.k16one, ref.k16[:, col].contiguous()):
                            fail(f"ld_triangle {path} {f1} against the {fmt} cells: {tag}")
        # launches chained by a HIP graph instead of a stream (round 5: such launches, into alternating result buffers, lost
        # their tickets): a few launches of either matrix kernel, sometimes forked onto a side stream, replayed twice
        if graphs < 2000 and rng.rand() < 0.05:
            gfmt = str(rng.choice(["k16", "ld32"]))
            gview = torch.int16 if gfmt == "k16" else torch.int32
            gref = ld_triangle(p, fmt=gfmt, path="popcount")
            bufs = [ld_triangle(p, fmt=gfmt, path="fp4"), ld_triangle(p, fmt=gfmt, path="fp4")]
            plan = [(str(rng.choice(["fp4", "mfma"])), int(k & 1), bool(rng.rand() < 0.3)) for k in range(int(rng.randint(2, 6)))]
            side = torch.cuda.Stream()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                cur = torch.cuda.current_stream()
                for gpath, b, fork in plan:
                    if fork:
                        side.wait_stream(cur)
                        with torch.cuda.stream(side):
                            ld_triangle(p, fmt=gfmt, path=gpath, out=bufs[b])
                        cur.wait_stream(side)
                    else:
                        ld_triangle(p, fmt=gfmt, path=gpath, out=bufs[b])
            for rep in range(2):
                for o in bufs:
                    o.cells.view(gview).fill_(-1)
                g.replay()
                torch.cuda.synchronize()
                for b in sorted({b for _, b, _ in plan}):
                    if not torch.equal(bufs[b].cells.view(gview), gref.cells.view(gview)):
                        fail(f"HIP graph of launches {plan} {gfmt}, buffer {b}, replay {rep}: {tag}")
            graphs += 1
            del g, bufs, gref
        # the other entry points on the same panel: explicit pair list, rectangular counts, the fused drop-in
        m = max(1, min(64, n * (n - 1) // 2))
        rr = rng.randint(1, n, size=m)
        cc = (rng.rand(m) * rr).astype(np.int64)
        full = ld_triangle(p, fmt="k16", path="fp4", want_n11=True)
        kk, int0, esc = full.k_and_int0(full.cell_index(rr, cc))
        lp = ld_pairs(p, rr, cc)
        k_exact = np.where((lp["flags"][:, None] & np.array([2, 1])) != 0, 0, lp["k"]).astype(np.float64)
        if not np.array_equal(np.where(esc, k_exact, kk), k_exact) or not np.array_equal(int0[:, 0], (lp["flags"] & 2) != 0) \
                or not np.array_equal(int0[:, 1], (lp["flags"] & 1) != 0):
            fail(f"ld_pairs vs triangle cells: {tag}")
        n11_cells = full.n11[torch.as_tensor(full.cell_index(rr, cc), device=full.n11.device)].cpu().numpy()
        if not np.array_equal(n11_cells.view(np.uint32), lp["n11"]):
            fail(f"ld_pairs n11 vs triangle n11 plane: {tag}")
        blk = pair_counts(p)[torch.as_tensor(rr, device=full.n11.device), torch.as_tensor(cc, device=full.n11.device)].cpu().numpy()
        if not np.array_equal(blk.view(np.uint32), lp["n11"]):
            fail(f"pair_counts vs ld_pairs: {tag}")
        host = codes[:, :h].cpu().numpy()
        for x in range(min(3, m)):
            res = calc_ld(host[rr[x]], host[cc[x]])
            want = {"r_square": 0 if lp["flags"][x] & 2 else lp["k"][x, 0] / 10000.0,
                    "d_prime": 0 if lp["flags"][x] & 1 else lp["k"][x, 1] / 10000.0}
            if str(res["r_square"]) != str(want["r_square"]) or str(res["d_prime"]) != str(want["d_prime"]):
                fail(f"drop-in calc_ld vs ld_pairs at ({rr[x]}, {cc[x]}): {res} {want}: {tag}")
        del full
        # ld_area
        pos = np.cumsum(rng.choice([0, 1, 37, 800, 5000], size=n, p=[0.03, 0.27, 0.4, 0.25, 0.05])) + 1
        flank = int(rng.choice([0, 500, 20000, 250000]))
        measure = str(rng.choice(["r_square", "d_prime"]))
        thres = float(rng.choice([0.0, 0.2, 0.8, 1.0]))
        queries = None if rng.rand() < 0.5 else sorted(set(rng.randint(0, n, size=int(rng.randint(1, n + 1))).tolist()))
        res = {}
        for path in ("popcount", "mfma", "fp4"):
            ops.set_area_path(path)
            res[path] = ld_area(p, pos, queries, flank, measure, thres)
        ops.set_area_path("auto")
        w = res["popcount"]
        for path in ("mfma", "fp4"):
            g = res[path]
            same = (len(g) == len(w) and g.n_pairs == w.n_pairs and torch.equal(g.query, w.query) and torch.equal(g.oppos, w.oppos)
                    and torch.equal(g.ld32.view(torch.int32), w.ld32.view(torch.int32)))
            if not same:
                fail(f"ld_area {path} flank={flank} {measure}>={thres} queries={'all' if queries is None else len(queries)}: {tag}")
        hits += len(w)
        rounds += 1
        del p, codes, res
        if progress and time.time() >= t_say:          # a long run must not look hung
            print(f"  ... {rounds} panels, no difference", flush=True)
            t_say = time.time() + progress
    return (f"fuzz ok: {rounds} panels, {pairs} triangle cells x 2 kernels x 2 launches, {hits} ld_area hits x 2 kernels, "
            f"{graphs} HIP graphs of launches, {budget:.0f} s")


if __name__ == "__main__":
    try:
        print(run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1, progress=30.0))
    except Mismatch as exc:
        print("FUZZ MISMATCH:", exc, flush=True)
        sys.exit(1)
