#!/usr/bin/env python3
"""Ad-hoc GPU measurements used while tuning (not part of the product or of bench.py's contract).

    python tools/gpu_exp.py counts     # n11-only kernel on a 10k x 10k block (counting without the epilogue)
    python tools/gpu_exp.py area       # configs[2]: 100k SNPs, +-500 kb windows, r2 >= 0.8
    python tools/gpu_exp.py tri100k    # configs[3] on one GPU: 100k x 5008 triangle
    python tools/gpu_exp.py eur        # configs[4] shape on the popcount path: 50k x 1008
    python tools/gpu_exp.py pack       # pack kernel bandwidth
    python tools/gpu_exp.py small      # configs[0] and other driver-sized tables: codes on the host -> ld_two_dim on the host
    python tools/gpu_exp.py calc       # the drop-in calc_ld pair by pair: microseconds per call
    python tools/gpu_exp.py nows       # ldx_triangle_ex_dev with and without the pass scheduler's workspace (round-robin passes)
"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402

from ld_tools_amd import PackedPanel, ld_area, ld_triangle, pair_counts, synth  # noqa: E402


def timed(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    what = sys.argv[1]
    out = {"exp": what}
    if what == "counts":
        n, h = 10000, 5008
        p = PackedPanel.from_codes(synth.synth_codes_device(n, h))
        ms = timed(lambda: pair_counts(p), reps=5)
        out.update(ms=ms, pairs=n * n, pairs_per_s=n * n / (ms * 1e-3), lane_ops_T=n * n * 314 / (ms * 1e-3) / 1e12)
    elif what == "small":
        import numpy as np

        from ld_tools_amd.drivers.ingest import k_to_python
        rows = []
        for n in (64, 256, 1000, 2000):
            codes = synth.synth_codes_host(n, 5008, seed=3)
            def whole():                                          # drivers/triangle.py: triangle_matrix without the VCF reads
                p = PackedPanel.from_codes(codes)                 # H2D + pack
                dense, fixes = ld_triangle(p).dense_values("r_square", None)
                flat = k_to_python(dense, fixes)
                return [flat[r * n:(r + 1) * n] for r in range(n)]   # the reference's ld_two_dim
            whole()
            t0 = time.perf_counter()
            for _ in range(3):
                whole()
            wall = (time.perf_counter() - t0) / 3
            p = PackedPanel.from_codes(codes)
            ms = timed(lambda: ld_triangle(p), reps=20)
            rows.append({"snps": n, "pairs": n * (n - 1) // 2, "kernel_ms": ms, "host_to_host_ms": wall * 1e3})
        out["rows"] = rows
    elif what == "area":
        n, h = 100000, 5008
        p = PackedPanel.from_codes(synth.synth_codes_device(n, h))
        pos = synth.synth_positions(n, step=500)
        t0 = time.perf_counter()
        hits = ld_area(p, pos, None, 500000, "r_square", 0.8)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        pos_d = torch.as_tensor(pos).to(p.device)                        # a driver keeps the chromosome's positions resident
        ld_area(p, pos_d, None, 500000, "r_square", 0.8, check_positions=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            hits = ld_area(p, pos_d, None, 500000, "r_square", 0.8, check_positions=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        out.update(first_call_s=first, s=dt, ordered_pairs=hits.n_pairs, hits=len(hits),
                   ordered_pairs_per_s=hits.n_pairs / dt)
    elif what == "area2":   # configs[2] as bench.py times it: the product call (one HIP graph) and the scan alone (HIP events)
        n, h = 100000, 5008
        p = PackedPanel.from_codes(synth.synth_codes_device(n, h))
        pos = torch.as_tensor(synth.synth_positions(n, step=500)).to(p.device)
        for _ in range(3):
            hits = ld_area(p, pos, None, 500000, "r_square", 0.8, check_positions=False)
        torch.cuda.synchronize()
        reps, ev, scan = 20, [], []
        t0 = time.perf_counter()
        for _ in range(reps):
            hits = ld_area(p, pos, None, 500000, "r_square", 0.8, check_positions=False)
            torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        for _ in range(reps):
            ld_area(p, pos, None, 500000, "r_square", 0.8, check_positions=False, events=ev)
            torch.cuda.synchronize()
            scan.append(ev[0].elapsed_time(ev[1]))
        scan.sort()
        out.update(end_to_end_ms=wall * 1e3, scan_ms_median=scan[len(scan) // 2], scan_ms_min=scan[0], hits=len(hits),
                   ordered_pairs=hits.n_pairs)
    elif what == "nows":   # round 6: ticket counter in the caller's workspace against round-robin passes (workspace = NULL)
        import statistics

        from ld_tools_amd import _lib
        from ld_tools_amd._lib import lib
        res = {}
        for n, h, reps in ((3000, 5008, 60), (10000, 5008, 60), (40000, 5008, 9), (50000, 1008, 9)):
            p = PackedPanel.from_codes(synth.synth_codes_device(n, h))
            o = ld_triangle(p, fmt="k16")
            st = torch.cuda.current_stream().cuda_stream

            def launch(ws):
                rc = lib.ldx_triangle_ex_dev(p.alt.data_ptr(), p.fa.data_ptr(), p.fr.data_ptr(), p.q.data_ptr(), p.n_snps, p.n_hap, 0,
                                             p.n_units, 3, _lib.FORMATS["k16"], o.cells.data_ptr(), None, None,
                                             o.ws.data_ptr() if ws else None, o.ws.numel() if ws else 0, st)
                assert rc == 0

            for _ in range(max(3, reps // 3)):
                launch(True)
            torch.cuda.synchronize()
            ms = {True: [], False: []}
            for _ in range(reps):          # interleaved
                for ws in (True, False):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    launch(ws)
                    b.record()
                    torch.cuda.synchronize()
                    ms[ws].append(a.elapsed_time(b))
            res[f"{n}x{h}"] = {"workspace_ms": statistics.median(ms[True]), "round_robin_ms": statistics.median(ms[False]),
                               "slower": statistics.median(ms[False]) / statistics.median(ms[True]) - 1.0}
            del o, p
            torch.cuda.empty_cache()
        out.update(res)
    elif what == "tri100k":
        n, h = 100000, 5008
        p = PackedPanel.from_codes(synth.synth_codes_device(n, h))
        res = ld_triangle(p)
        ms = timed(lambda: ld_triangle(p, out=res), reps=3)
        out.update(ms=ms, pairs=p.n_pairs, pairs_per_s=p.n_pairs / (ms * 1e-3), out_gb=res.ld32.numel() * 4 / 1e9)
    elif what == "eur":
        n, h = 50000, 1008
        p = PackedPanel.from_codes(synth.synth_codes_device(n, h))
        res = ld_triangle(p)
        ms = timed(lambda: ld_triangle(p, out=res), reps=3)
        out.update(ms=ms, pairs=p.n_pairs, pairs_per_s=p.n_pairs / (ms * 1e-3))
    elif what == "pack":
        n, h = 100000, 5008
        codes = synth.synth_codes_device(n, h)
        p = PackedPanel.empty(n, h)
        ms = timed(lambda: p.pack_from(codes), reps=5)
        out.update(ms=ms, gbps_in=n * h / (ms * 1e-3) / 1e9)
    elif what == "calc":
        # the drop-in calc_ld, pair by pair (INTEGRATION.md level 0): microseconds per call for lists and for int8 arrays
        import numpy as np

        from ld_tools_amd.backend.calc_ld import calc_ld
        codes = synth.synth_codes_host(2, 5008, seed=3)
        g1, g2 = codes[0].tolist(), codes[1].tolist()
        a1, a2 = codes[0].copy(), codes[1].copy()
        for name, x, y in (("lists", g1, g2), ("int8_arrays", a1, a2)):
            calc_ld(x, y)
            t0 = time.perf_counter()
            for _ in range(300):
                calc_ld(x, y)
            out[name + "_us_per_call"] = (time.perf_counter() - t0) / 300 * 1e6
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
