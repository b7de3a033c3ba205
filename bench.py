#!/usr/bin/env python3
"""Benchmark of the pairwise-LD hot path on MI355X: SNP-pairs/s producing r^2 + D'.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--snps S] [--haps H] [--fmt k16|ld32] [--path fp4|mfma|popcount]

One "step" = one full pass of the hot path over one synthetic panel that is already resident in HBM as packed slab
shards: (N > 1: all-gather of the shards over RCCL) -> ld_triangle kernel over this rank's share of the pass list ->
one result cell per pair written to HBM (4 bytes: k = round(x, 4) * 10^4 for r^2 and D' as two uint16 -- lossless with
respect to the reference's rounded output; --fmt ld32 writes the 8-byte float32 pair instead).

Workloads (BASELINE.json): N = 1 -> configs[1], ld_triangle 10 000 SNPs x 5008 haplotypes.  N > 1 -> configs[3],
100 000 x 5008, a FIXED problem whose pass list is cut into N equal contiguous ranges (strong scaling): row-block
shards packed per rank, exchanged by ONE all-gather per step, every rank computes its range against the full plane
(ld_tools_amd/dist.py).  --snps overrides either.

Launch: `python bench.py --gpus N` with N > 1 starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
itself, as a CHILD process and before this process has touched the GPU, relays rank 0's JSON line and exits non-zero
if the child fails or overruns --deadline; under torch.distributed.run (RANK / WORLD_SIZE set) it runs as a rank.

Timing: W warm-up steps, then --settle-steps more untimed steps (the shader clock needs tens of ms of load to settle;
reported as config.settle_steps, and the from-idle figure is reported beside it as cold_ms_per_step), then EXACTLY K
steps between barrier + synchronize on both sides, max over ranks.  At N = 1 the K steps are replayed as one HIP graph
(eager with --no-graph; a failed capture falls back to eager): a 0.13 ms kernel is launch-bound otherwise.  At N > 1 a
step is milliseconds and the launches are eager unless --graph-dist asks for a captured RCCL exchange; a failed child
run is repeated once in the plainest mode (eager, exchange not overlapped).  The output is poisoned before the timed
region and compared bit for bit with a separately computed result after it.

Prints ONE JSON line on rank 0: whole-job pairs/s, ms per step, the roofline object of the dominant kernel (HIP events
on the launch stream, live), at N = 1 the other two kernel paths on the same workload (int8 MFMA, AND+popcount) and the
CPU baseline (the pure-Python restatement of the reference's list/zip/count algorithm, 1 core, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_FP4_PEAK_TOPS = 10000.0              # ... FP4 / FP6 MFMA dense = 4x the BF16 rate per clock = ~10 POP/s (no sparsity)
MFMA_I8_PEAK_TOPS = 5000.0                # ... I8 MFMA dense = 2x BF16 = ~5 POP/s
VALU_PEAK_TLANEOPS = 256 * 64 * 2.4e9 / 1e12   # AND / BCNT have no packed form: 64 lanes/clk/CU at 2.4 GHz = 39.3 T


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=7,
                    help="timed regions of --steps steps each; the line reports the median region (and all of them)")
    ap.add_argument("--snps", type=int, default=0, help="panel size (default: 10 000 at N = 1, 100 000 at N > 1)")
    ap.add_argument("--haps", type=int, default=5008)
    ap.add_argument("--fmt", default="k16", choices=("k16", "ld32"),
                    help="result cells: k16 = 4 bytes per pair (two uint16 k = value * 10^4, lossless), ld32 = 8 bytes (two float32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="N = 1: skip the int8 / popcount / other-format legs")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="N = 1: skip BASELINE configs[4] (50 000 x 1008 triangle), configs[2] (100 000-SNP ld_area) and the "
                         "pack / host-to-device timings that follow the headline")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--force-dist", action="store_true", help="run the sharded path (process group + all-gather) even at world 1")
    ap.add_argument("--cpu-sample-snps", type=int, default=320)
    ap.add_argument("--no-graph", action="store_true", help="launch the steps eagerly instead of replaying a HIP graph")
    ap.add_argument("--graph-dist", action="store_true",
                    help="N > 1: capture the K steps, RCCL exchange included, into one HIP graph (default at N > 1: eager)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: finish each step's all-gather before its kernel instead of overlapping it with the "
                         "previous step's kernel")
    ap.add_argument("--overlap", action="store_true", help="use the overlapped exchange even in a one-rank group (rehearsals)")
    ap.add_argument("--settle-steps", type=int, default=-1,
                    help="untimed steps run right before the timed region, on top of --warmup, so that it starts at "
                         "sustained clocks (default: ~0.15 s worth; 0 = off)")
    ap.add_argument("--path", default="auto", choices=("auto", "fp4", "mfma", "popcount"),
                    help="kernel behind ld_triangle (auto = the FP4 MFMA kernel; results are identical)")
    ap.add_argument("--deadline", type=float, default=1500.0, help="seconds the launcher waits for the rank processes")
    ap.add_argument("--pg-timeout", type=float, default=300.0, help="seconds a collective may take before the job aborts")
    ap.add_argument("--debug-hang", action="store_true",
                    help="test hook: every rank sleeps instead of running (the launcher must end the whole process group at "
                         "--deadline and return 124: tests/test_dist_gloo.py)")
    ap.add_argument("--debug-corrupt-result", action="store_true",
                    help="test hook: rank 0 flips one result cell before the check of the timed region (the run must then "
                         "end with exit code 97, unretried: tests/test_gpu_dist.py)")
    ap.add_argument("--other-scale", default="full", choices=("full", "small"),
                    help="tests: 'small' shrinks the other_workloads panels (6 000 x 1008, 12 000 x 5008)")
    ap.add_argument("--debug-corrupt-other", action="store_true",
                    help="test hook: one cell of the configs[4] leg of other_workloads is flipped before its check (the run "
                         "must end with exit code 97 and no throughput line: tests/test_gpu_dist.py)")
    ap.add_argument("--no-single-gpu-leg", action="store_true",
                    help="N > 1: skip timing the whole workload on rank 0 alone (the strong-scaling reference)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ launcher (no GPU use)
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


RC_VERIFY = 97      # a rank's timed steps did not reproduce the triangle: a wrong RESULT, never retried, never hidden
VERIFY_MARK = "BENCH_VERIFY_FAILED"   # ... and the line the rank prints on stdout: torch.distributed.run turns every worker
                                      # failure into its own exit code 1, so the launcher recognises the case by this line
RC_DEADLINE = 124   # the rank processes overran --deadline


def launch_ranks(args, argv) -> int:
    """Start one rank per GPU through torch.distributed.run in a child process and relay rank 0's JSON line.  This
    process never initialises HIP (an exec or fork after HIP init is what must not happen; a plain child is fine).

    The child runs in its own session: on a deadline overrun the WHOLE process group (torch.distributed.run and the rank
    grandchildren that hold the GPUs) is terminated, and nothing is started beside it afterwards.  A failed run is
    repeated once in the plainest mode (eager launches, exchange not overlapped) -- except after a verification failure
    (RC_VERIFY) or a deadline overrun, which are returned as they are; the JSON line says whether it is a second attempt."""
    import signal

    def attempt(extra):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve())] + argv + extra
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, start_new_session=True)
        try:
            out, _ = proc.communicate(timeout=args.deadline)
            return proc.returncode, out or ""
        except subprocess.TimeoutExpired:
            print(f"[bench] the rank processes overran the deadline of {args.deadline:.0f} s: ending their process group",
                  file=sys.stderr)
            for sig, grace in ((signal.SIGTERM, 15.0), (signal.SIGKILL, 15.0)):
                try:
                    os.killpg(proc.pid, sig)        # the session leader's pid is the group id
                except ProcessLookupError:
                    break
                try:
                    proc.wait(timeout=grace)
                    break
                except subprocess.TimeoutExpired:
                    continue
            try:        # torch.distributed.run may be gone while a rank grandchild (blocked in a driver call) is not:
                os.killpg(proc.pid, signal.SIGKILL)     # the group, not the leader, is what must be gone
            except ProcessLookupError:
                pass
            try:
                out, _ = proc.communicate(timeout=5.0)
            except Exception:                       # noqa: BLE001
                out = ""
            return RC_DEADLINE, out or ""

    def json_line(text):
        for line in reversed(text.splitlines()):
            if line.startswith("{") and '"metric"' in line:
                return line
        return None

    rc, out = attempt([])
    line = json_line(out)
    first_rc, retried = rc, False
    if rc == RC_VERIFY or VERIFY_MARK in out:
        print("[bench] a rank's timed steps did not reproduce the triangle: NOT retried", file=sys.stderr)
        sys.stdout.write(out)
        return RC_VERIFY
    if rc == RC_DEADLINE:
        sys.stdout.write(out)
        return RC_DEADLINE
    if (rc != 0 or line is None) and not (args.no_graph and args.no_overlap):
        print(f"[bench] rank processes failed (rc {rc}); once more in the plainest mode: eager launches, exchange not "
              "overlapped", file=sys.stderr)
        retried = True
        rc, out = attempt(["--no-graph", "--no-overlap"])
        line = json_line(out)
    if line is None:
        sys.stdout.write(out)
        return rc or 1
    try:        # the record says how it came about
        rec = json.loads(line)
        rec["retried"] = retried
        rec["first_attempt_rc"] = first_rc
        rec["first_attempt_mode"] = "as requested" if not retried else "as requested (failed); this line: --no-graph --no-overlap"
        line = json.dumps(rec)
    except ValueError:
        pass
    print(line, flush=True)
    return rc


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(codes_host, sample_snps):
    """Time the oracle's pure-Python list/zip/count path (kind "port") on the first rows of the bench panel."""
    from oracle import c_oracle
    from oracle import ld_oracle as orc

    rows = [r.tolist() for r in codes_host[:sample_snps]]
    t0 = time.perf_counter()
    pairs = 0
    for i in range(len(rows)):
        gi = rows[i]
        for j in range(i):
            orc.calc_ld_lists(gi, rows[j])
            pairs += 1
    dt = time.perf_counter() - t0
    out = {"value": pairs / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
           "sample": f"all {pairs} row>col pairs of the first {len(rows)} SNPs of the bench panel "
                     f"({codes_host.shape[1]} haplotypes), oracle/ld_oracle.py calc_ld_lists, {dt:.1f} s",
           "cpu": _cpu_model(), "host_cores": os.cpu_count()}
    # the same sample over a process pool capped like the reference's (min(8, cores), ld_triangle.py:394-399); the
    # workers are spawned (this process has initialised HIP and must not fork) and import the oracle only
    try:
        import multiprocessing as mp
        nproc = min(8, os.cpu_count() or 1)
        if nproc > 1:
            n = len(rows)
            cuts = [int(round(n * math.sqrt(k / nproc))) for k in range(nproc + 1)]    # equal pair counts per worker
            with mp.get_context("spawn").Pool(nproc) as pool:          # every wait is bounded: an extra must not hang the bench
                pool.starmap_async(orc.triangle_rows_lists, [(rows[:2], 0, 2)] * nproc).get(timeout=120)   # workers up
                t0 = time.perf_counter()
                done = sum(pool.starmap_async(orc.triangle_rows_lists,
                                              [(rows, cuts[k], cuts[k + 1]) for k in range(nproc)]).get(timeout=180))
                dtp = time.perf_counter() - t0
            out["pool"] = {"value": done / dtp, "unit": "pairs/s", "cores": nproc,
                           "sample": f"the same {done} pairs over multiprocessing.Pool({nproc}), spawn, {dtp:.1f} s "
                                     "(argument pickling included)"}
    except Exception as exc:   # noqa: BLE001  (a reported extra, never a reason to lose the bench line)
        out["pool"] = {"error": f"{type(exc).__name__}: {exc}"}
    # the C restatement (AND + popcount + fp64 mirror), one thread, as the stronger CPU comparator
    n_c = min(codes_host.shape[0], 1536)
    p = c_oracle.Panel(codes_host[:n_c])
    t0 = time.perf_counter()
    p.triangle(want=("rsq_rnd", "dp_rnd"))
    dtc = time.perf_counter() - t0
    out["native"] = {"value": (n_c * (n_c - 1) // 2) / dtc, "unit": "pairs/s", "cores": 1,
                     "sample": f"{n_c}-SNP triangle, oracle/ld_oracle.c, {dtc:.1f} s"}
    return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ------------------------------------------------------------------------------------------ other workloads (N = 1)
def other_workloads(torch, dev, fmt, bench_codes, scale="full", corrupt=False):
    """BASELINE.json configs[4], configs[3]'s panel on ONE GPU and configs[2] on the driver-run line, each a few launches
    at settled clocks and each checked against a second, independently computed result (a mismatch ends the run with
    RC_VERIFY, like a mismatch of the headline); plus what SURVEY 8d asks to report beside pairs/s: the pack kernel and
    the host-to-device copy of the bench panel's codes (never part of `value`).  `scale` = "small" shrinks the three
    panels (tests only; the keys then carry the sizes actually run)."""
    from ld_tools_amd import PackedPanel, dist as ldist, ld_area, ld_triangle, ops, synth
    from ld_tools_amd._lib import UNIT_PAIRS, lib

    def events():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    small = scale != "full"
    res = {}
    cell_bytes = 4 if fmt == "k16" else 8
    import statistics

    def timed_launches(fn, reps):
        """`reps` launches of fn() back to back, HIP events around each on the launch stream: (median, min, all) in ms."""
        evs = []
        for _ in range(reps):
            a, c = events()
            a.record()
            fn()
            c.record()
            evs.append((a, c))
        torch.cuda.synchronize()
        ms = [a.elapsed_time(c) for a, c in evs]
        return statistics.median(ms), min(ms), ms
    # ---- configs[4]: ld_triangle 50 000 x 1008 (EUR sub-panel), the HBM-write regime ----
    n, h = (6000, 1008) if small else (50000, 1008)
    key4 = f"ld_triangle {n}x{h}"
    try:
        p = PackedPanel.from_codes(synth.synth_codes_device(n, h, seed=synth.BENCH_SEED, device=dev))
        o = ld_triangle(p, fmt=fmt)
        for _ in range(3 if small else 40):     # untimed: ~60 ms of load, so that the timed launches run at settled clocks
            ld_triangle(p, out=o, fmt=fmt)      # (the headline's settle_steps, for this leg; reported as settle_launches)
        torch.cuda.synchronize()
        o.cells.fill_(-1)
        reps = 9
        ms, ms_min, ms_all = timed_launches(lambda: ld_triangle(p, out=o, fmt=fmt), reps)
        # the same panel through the one-measure path (2 B/pair: what ld_triangle's table writer asks for, one measure per
        # run; never the headline: the metric is r2 + D'); every cell against the r2 half of the two-value result
        one = ld_triangle(p, fmt="k16r") if fmt == "k16" else None
        if one is not None:
            for _ in range(3 if small else 10):
                ld_triangle(p, out=one, fmt="k16r")
            torch.cuda.synchronize()
            one.cells.fill_(-1)
            ms1, ms1_min, ms1_all = timed_launches(lambda: ld_triangle(p, out=one, fmt="k16r"), reps)
            same_one = bool(torch.equal(one.k16one, o.k16[:, 0].contiguous()))
        if corrupt:                                              # test hook (--debug-corrupt-other)
            o.cells.view(torch.int32).view(-1)[4321] ^= 1
        chk = ld_triangle(p, fmt=fmt, path="popcount")          # the independent kernel (AND + popcount, fp64 epilogue)
        torch.cuda.synchronize()
        same = bool(torch.equal(chk.cells.view(torch.int32), o.cells.view(torch.int32)))
        del chk
        pairs = n * (n - 1) // 2
        alg = float(cell_bytes) * pairs + lib.ldx_plane_bytes(n, h)
        res[key4] = {
            "ms": ms, "ms_min": ms_min, "ms_runs": ms_all, "timing": "median of the launches (HIP events around each)",
            "pairs_per_s": pairs / (ms * 1e-3), "fmt": fmt, "verified_against": "popcount kernel, every cell",
            "results_equal": same, "launches": reps, "settle_launches": 3 if small else 40,
            "roofline_hbm": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg},
            "frac_of_fp4_peak": 2.0 * h * pairs / (ms * 1e-3) / 1e12 / MFMA_FP4_PEAK_TOPS}
        if one is not None:
            alg1 = 2.0 * pairs + lib.ldx_plane_bytes(n, h)
            res[key4 + ", r2 only (2 B/pair)"] = {
                "ms": ms1, "ms_min": ms1_min, "ms_runs": ms1_all, "pairs_per_s": pairs / (ms1 * 1e-3), "fmt": "k16r",
                "what": "LDX_OUT_K16_RSQ: one measure per pair, the other value's arithmetic skipped (the table writer's path)",
                "verified_against": "the r2 half of every cell of the two-value result above", "results_equal": same_one,
                "launches": reps,
                "roofline_hbm": {"bound": "hbm", "achieved": alg1 / (ms1 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": alg1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg1}}
            del one
        del o, p
        torch.cuda.empty_cache()
    except Exception as exc:   # noqa: BLE001  (a reported extra, never a reason to lose the bench line)
        res[key4] = {"error": f"{type(exc).__name__}: {exc}"}
    # ---- the 100 000 x 5008 panel: configs[3]'s triangle on ONE GPU (the 1-GPU point of the strong-scaling series and
    #      the panel the >= 40 % target is written for), then configs[2]'s ld_area on the same resident panel ----
    n, h = (12000, 5008) if small else (100000, 5008)
    key3, key2 = f"ld_triangle {n}x{h}", f"ld_area {n} +-500kb r2>=0.8"
    p = None
    try:
        p = PackedPanel.from_codes(synth.synth_codes_device(n, h, seed=synth.BENCH_SEED, device=dev))
    except Exception as exc:   # noqa: BLE001
        res[key3] = res[key2] = {"error": f"{type(exc).__name__}: {exc}"}
    if p is not None:
        try:
            o = ld_triangle(p, fmt=fmt)                              # 20 GB of 4-byte cells at 100 000 SNPs
            ld_triangle(p, out=o, fmt=fmt)
            torch.cuda.synchronize()
            o.cells.fill_(-1)
            reps = 5
            t0 = time.perf_counter()
            kern_ms, kern_min, kern_all = timed_launches(lambda: ld_triangle(p, out=o, fmt=fmt), reps)
            ms = kern_ms            # (a launch is ~10 ms: the events around it are the step)
            # verification: whole unit ranges of the eight-way partition, recomputed by the popcount kernel
            parts = ldist.unit_partition(n, 8)
            checked, same = [], True
            for r in (2, 7):
                u0, u1 = parts[r]
                chk = ld_triangle(p, unit_range=(u0, u1), fmt=fmt, path="popcount")
                torch.cuda.synchronize()
                mine = o.cells.view(torch.int32)[u0 * UNIT_PAIRS:u1 * UNIT_PAIRS]
                same = same and bool(torch.equal(chk.cells.view(torch.int32), mine))
                checked.append(r)
                del chk, mine
            pairs = n * (n - 1) // 2
            alg_ops = 2.0 * h * pairs
            alg = float(cell_bytes) * pairs + lib.ldx_plane_bytes(n, h)
            res[key3] = {
                "ms": ms, "ms_min": kern_min, "ms_runs": kern_all, "timing": "median of the launches (HIP events around each)",
                "pairs_per_s": pairs / (ms * 1e-3), "fmt": fmt, "launches": reps,
                "verified_against": f"popcount kernel, every cell of unit ranges {checked} of unit_partition({n}, 8)",
                "results_equal": same,
                "roofline": {"bound": "mfma", "achieved": alg_ops / (kern_ms * 1e-3) / 1e12, "peak": MFMA_FP4_PEAK_TOPS,
                             "unit": "TOP/s", "frac": alg_ops / (kern_ms * 1e-3) / 1e12 / MFMA_FP4_PEAK_TOPS,
                             "kernel": "triangle_mfma_kernel", "kernel_ms": kern_ms, "algorithmic_ops": alg_ops},
                "roofline_hbm": {"bound": "hbm", "achieved": alg / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "algorithmic_bytes": alg}}
            del o
            torch.cuda.empty_cache()
        except Exception as exc:   # noqa: BLE001
            res[key3] = {"error": f"{type(exc).__name__}: {exc}"}
            torch.cuda.empty_cache()
        # ---- configs[2]: ld_area, 100 000 SNPs, +-500 kb, r2 >= 0.8, every SNP a query ----
        try:
            pos = torch.as_tensor(synth.synth_positions(n, step=500)).to(dev)
            hits = ld_area(p, pos, None, 500000, "r_square", 0.8)
            for _ in range(2):
                ld_area(p, pos, None, 500000, "r_square", 0.8, check_positions=False)
            torch.cuda.synchronize()
            reps, ev, walls, scans = 7, [], [], []
            for _ in range(reps):       # the product call: from its second repetition on the launches are one HIP graph
                t0 = time.perf_counter()
                hits = ld_area(p, pos, None, 500000, "r_square", 0.8, check_positions=False)
                torch.cuda.synchronize()
                walls.append(time.perf_counter() - t0)
            for _ in range(reps):       # the scan alone: HIP events need the launches issued one by one
                ld_area(p, pos, None, 500000, "r_square", 0.8, check_positions=False, events=ev)
                torch.cuda.synchronize()
                scans.append(ev[0].elapsed_time(ev[1]))
            wall, scan_ms = statistics.median(walls), statistics.median(scans)
            old = ops.get_area_path()
            try:
                ops.set_area_path("popcount")                        # the independent scan kernel
                ref = ld_area(p, pos, None, 500000, "r_square", 0.8, check_positions=False)
            finally:
                ops.set_area_path(old)
            same = bool(len(ref) == len(hits) and torch.equal(ref.query, hits.query) and torch.equal(ref.oppos, hits.oppos)
                        and torch.equal(ref.ld32.view(torch.int32), hits.ld32.view(torch.int32)))
            n_pairs = hits.n_pairs
            res[key2] = {
                "end_to_end_ms": wall * 1e3, "ordered_pairs": n_pairs, "ordered_pairs_per_s": n_pairs / wall, "hits": len(hits),
                "scan_ms": scan_ms, "scan": "ldx_area_scan_dev: query mask, band plan, FP4 band kernel (HIP events, eager launches)",
                "end_to_end_ms_min": min(walls) * 1e3, "end_to_end_ms_runs": [w * 1e3 for w in walls], "scan_ms_min": min(scans),
                "scan_ms_runs": scans, "timing": f"medians of {reps} calls each",
                "end_to_end": "positions resident on the device; scan (counting per query) + offsets / scatter / order kernels replayed "
                              "as one HIP graph + one host read",
                "verified_against": "popcount scan, every hit in order", "results_equal": same}
            del hits, ref
        except Exception as exc:   # noqa: BLE001
            res[key2] = {"error": f"{type(exc).__name__}: {exc}"}
        del p
        torch.cuda.empty_cache()
    # ---- panels that are NOT all "ordinary" (VERDICT r05 item 2), each timed INTERLEAVED with its clean twin (same seed, same
    #      box, same minute) and checked against the popcount kernel: (i) configs[4]'s shape with 30 % of the SNPs
    #      monomorphic -- what an EUR-size sub-panel of the ALL-panel variants an ld_area window returns looks like
    #      (ld_area.py:215-225; the reference maps such pairs to the int 0, calc_ld.py:55-76,89-90); (ii) 40 000 x 5008 with
    #      0.1 % code 2 in 20 % of the rows (a + r < n).  Rounds 1-5 measured only panels with 0.07 % such SNPs. ----
    odd_legs = [((6000, 1008) if small else (50000, 1008), "30 % of rows monomorphic", dict(mono=0.3)),
                ((5000, 5008) if small else (40000, 5008), "0.1 % code-2 in 20 % of rows", dict(miss=0.001, miss_rows=0.2))]
    for (n, h), what, kw in odd_legs:
        key = f"ld_triangle {n}x{h}, {what}"
        try:
            pc = PackedPanel.from_codes(synth.synth_codes_device(n, h, seed=synth.BENCH_SEED, device=dev))
            po = PackedPanel.from_codes(synth.synth_codes_device(n, h, seed=synth.BENCH_SEED, device=dev, **kw))
            oc, oo = ld_triangle(pc, fmt=fmt), ld_triangle(po, fmt=fmt)
            for _ in range(3 if small else 12):
                ld_triangle(pc, out=oc, fmt=fmt)
                ld_triangle(po, out=oo, fmt=fmt)
            torch.cuda.synchronize()
            oo.cells.fill_(-1)
            reps, evc, evo = 7, [], []
            for _ in range(reps):
                for pp, out_, lst in ((pc, oc, evc), (po, oo, evo)):
                    a, c = events()
                    a.record()
                    ld_triangle(pp, out=out_, fmt=fmt)
                    c.record()
                    lst.append((a, c))
            torch.cuda.synchronize()
            msc = [a.elapsed_time(c) for a, c in evc]
            mso = [a.elapsed_time(c) for a, c in evo]
            chk = ld_triangle(po, fmt=fmt, path="popcount")
            torch.cuda.synchronize()
            same = bool(torch.equal(chk.cells.view(torch.int32), oo.cells.view(torch.int32)))
            del chk
            pairs = n * (n - 1) // 2
            m_c, m_o = statistics.median(msc), statistics.median(mso)
            res[key] = {"ms": m_o, "ms_min": min(mso), "ms_runs": mso, "pairs_per_s": pairs / (m_o * 1e-3),
                        "clean_twin_ms": m_c, "clean_twin_ms_runs": msc, "slower_than_clean_twin": m_o / m_c - 1.0,
                        "non_ordinary_snps": {"acnt_zero_or_full": int(((po.acnt[:n] == 0) | (po.acnt[:n] == h)).sum().item()),
                                              "with_missing_codes": int(((po.acnt[:n] + po.rcnt[:n]) < h).sum().item())},
                        "fmt": fmt, "launches": reps, "timing": "clean twin and this panel launched alternately; medians",
                        "verified_against": "popcount kernel, every cell", "results_equal": same}
            del pc, po, oc, oo
            torch.cuda.empty_cache()
        except Exception as exc:   # noqa: BLE001
            res[key] = {"error": f"{type(exc).__name__}: {exc}"}
            torch.cuda.empty_cache()
    # ---- pack and host-to-device of the bench panel's codes (reported separately, not in pairs/s) ----
    try:
        ns, nh = bench_codes.shape
        pk = PackedPanel.empty(ns, nh, device=dev)
        pk.pack_from(bench_codes)
        torch.cuda.synchronize()
        a, c = events()
        a.record()
        for _ in range(20):
            pk.pack_from(bench_codes)
        c.record()
        torch.cuda.synchronize()
        pack_ms = a.elapsed_time(c) / 20
        host = bench_codes.cpu().pin_memory()
        dst = torch.empty_like(bench_codes)
        dst.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        a, c = events()
        a.record()
        for _ in range(5):
            dst.copy_(host, non_blocking=True)
        c.record()
        torch.cuda.synchronize()
        h2d_ms = a.elapsed_time(c) / 5
        res["ingest"] = {"pack_ms": pack_ms, "h2d_ms": h2d_ms, "codes_bytes": int(bench_codes.numel()),
                         "pack_gb_per_s_in": bench_codes.numel() / (pack_ms * 1e-3) / 1e9,
                         "h2d_gb_per_s": bench_codes.numel() / (h2d_ms * 1e-3) / 1e9,
                         "note": f"int8 codes [{ns}][{nh}] of the bench panel: pinned host -> device, then pack_codes + "
                                 "snp_stats on the device; outside the timed region (the panel stays resident across steps)"}
    except Exception as exc:   # noqa: BLE001
        res["ingest"] = {"error": f"{type(exc).__name__}: {exc}"}
    return res


# ------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    import datetime

    if args.debug_hang:          # before anything touches the GPU: the launcher's deadline handling is testable on any box
        print(f"[bench] rank {os.environ.get('RANK', '0')} pid {os.getpid()} hangs on request", file=sys.stderr, flush=True)
        time.sleep(3600)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU path")
    if args.backend != "nccl":
        local_rank %= torch.cuda.device_count()       # gloo rehearsal: several ranks may share one card
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if "RANK" not in os.environ:                  # --force-dist without a launcher: a one-rank group
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        tmo = datetime.timedelta(seconds=args.pg_timeout)     # a hung exchange aborts the job instead of hanging it
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo)

    from ld_tools_amd import PackedPanel, dist as ldist, ld_triangle, ops, synth
    from ld_tools_amd._lib import lib

    ops.set_triangle_path(args.path)
    path = "fp4" if args.path == "auto" else args.path
    fmt = args.fmt
    cell_bytes = 4 if fmt == "k16" else 8
    n_hap = args.haps
    n_snps = args.snps or (10000 if world == 1 else 100000)        # BASELINE.json configs[1] / configs[3]
    n_pairs = n_snps * (n_snps - 1) // 2

    # ---- setup (untimed): every rank ingests and packs its own row block ----
    b, e = ldist.slab_partition(n_snps, world)[rank]
    local = None
    codes_local = None
    if e > b:
        codes_local = synth.synth_codes_device(e - b, n_hap, seed=synth.BENCH_SEED, snp_offset=b, device=dev)
        local = PackedPanel.from_codes(codes_local)
    u0, u1 = ldist.unit_partition(n_snps, world)[rank]
    out = None
    panel = None if use_dist else local
    ev_k0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev_k1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    # N > 1: every step exchanges the packed shards (ONE RCCL all-gather) and runs this rank's share of the kernel.
    # The exchange of step k + 1 is issued before the kernel of step k and finished after it (double-buffered
    # panels, dist.PanelPipeline): RCCL works on its own stream, so the all-gather over xGMI rides under the
    # kernel instead of in front of it.  run_steps(K) = K exchanges + K kernels, nothing left in flight.
    overlap = use_dist and not args.no_overlap and (world > 1 or args.overlap)   # a one-rank group has nothing to hide
    pipe = ldist.PanelPipeline(n_snps, n_hap, dev) if overlap else None

    def step(k=None, first=True, last=True):
        nonlocal out, panel
        if pipe is not None:
            if first:
                pipe.start(local)
            panel = pipe.finish()
            if not last:
                pipe.start(local)
        elif use_dist:
            panel = ldist.all_gather_panel(local, n_snps, n_hap, out=panel)
        if k is not None:
            ev_k0[k].record()
        out = ld_triangle(panel, unit_range=(u0, u1), out=out, fmt=fmt)
        if k is not None:
            ev_k1[k].record()

    def run_steps(count, timed=False):
        for k in range(count):
            step(k if timed else None, first=(k == 0), last=(k == count - 1))

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(fn):
        fence()
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        fn()
        c.record()
        fence()
        return time.perf_counter() - t0, a.elapsed_time(c)

    run_steps(args.warmup)
    fence()
    # from-idle figure: the K steps right after the warm-up, before any clock settling (what a short job sees)
    cold_dt, _ = timed_region(lambda: run_steps(args.steps))
    # The K steps are K back-to-back launches of a ~0.15 ms kernel (plus, for N > 1, the exchange's all-gather and a
    # handful of small copies): a launch-bound inner loop, captured once into a HIP graph and replayed inside the
    # timed region; any failure to capture falls back to eager launches.
    graph = None
    # N = 1: always (a 0.13 ms kernel is launch-bound).  N > 1: a step is milliseconds, eager launches cost nothing, and a
    # captured RCCL exchange is one more thing that can go wrong on a node this code has never met: only on request.
    if not args.no_graph and (not use_dist or (args.graph_dist and args.backend == "nccl")):
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run_steps(args.steps)
            g.replay()                      # one untimed replay (also proves the graph runs)
            torch.cuda.synchronize()
            graph = g
        except Exception as exc:            # noqa: BLE001
            print(f"[bench] HIP graph capture unavailable ({type(exc).__name__}: {exc}); eager launches", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    # Clock settling (untimed, reported in config): a 20-step timed region is a few ms, far shorter than the tens of ms
    # the shader clock takes to settle under load.  A step COUNT, so every rank issues the same collectives.
    settle = args.settle_steps
    if settle < 0:
        settle = max(args.steps, int(0.15 / max(cold_dt / args.steps, 1e-6)))
        settle = min(settle, 2000)
        if use_dist:                        # every rank must run the same number of collectives
            t = torch.tensor([settle], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            settle = int(t.item())
    settle_done = 0
    if settle > 0:
        if graph is not None:
            for _ in range((settle + args.steps - 1) // args.steps):
                graph.replay()
                settle_done += args.steps
        else:
            run_steps(settle)
            settle_done = settle
    if out is not None:
        out.cells.fill_(-1)                # the timed steps must produce every result again (checked below)
    # "3 warm-ups, 10 timed, report median and min".  The timed region -- EXACTLY K steps between barrier +
    # synchronize on both sides -- is repeated R = 7 times (the same graph; the output was poisoned once, before the first);
    # `ms_per_step`, `value` and `roofline.frac` come from the MEDIAN region, all R are printed (ms_per_step_runs) with their
    # minimum beside them: one 2.5-ms region on a pool whose boxes differ by 8 % is not a measurement.
    repeats = max(1, args.repeats)
    runs = []           # (host seconds of the region, event span ms per step)
    for _ in range(repeats):
        if graph is not None:
            dt_r, span_ms = timed_region(graph.replay)
            km = span_ms / args.steps
        else:
            dt_r, _ = timed_region(lambda: run_steps(args.steps, timed=True))
            km = sum(a.elapsed_time(b_) for a, b_ in zip(ev_k0, ev_k1)) / args.steps
        runs.append((dt_r, km))

    def allmax(x):
        if not use_dist or world == 1:
            return x, x
        t = torch.tensor([x, -x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0].item()), -float(t[1].item())

    # max over ranks PER REGION (a region ends when its slowest rank does), then the median region
    per_region = [allmax(r[0])[0] for r in runs]
    order = sorted(range(repeats), key=lambda i: per_region[i])
    mid = order[(repeats - 1) // 2]                      # the median region (the lower one of an even count)
    dt = per_region[mid]
    dt_min = per_region[order[0]]
    kern_ms = runs[mid][1]
    cold_dt, _ = allmax(cold_dt)
    kern_ms_max, kern_ms_min = allmax(kern_ms)

    # the output of the timed region against a separately computed result -- by the INDEPENDENT kernel (AND + popcount counts,
    # fp64 epilogue; the FP4 kernel when popcount itself is timed): nothing was skipped, left stale, or computed wrongly
    check_path = "popcount" if path != "popcount" else "fp4"
    check = ld_triangle(panel, unit_range=(u0, u1), fmt=fmt, path=check_path)
    torch.cuda.synchronize()
    if args.debug_corrupt_result and rank == 0:
        out.cells.view(torch.int32).view(-1)[12345] ^= 1
    if not torch.equal(check.cells.view(torch.int32), out.cells.view(torch.int32)):
        print("bench.py: the timed steps did not reproduce the triangle (stale or skipped work)", file=sys.stderr, flush=True)
        print(f"{VERIFY_MARK} rank {rank}", flush=True)
        os._exit(RC_VERIFY)     # a distinct code the launcher propagates and never retries
    del check

    value = n_pairs * args.steps / dt
    # which devices did the ranks really run on?  (uuid where the runtime has one, else the PCI address)
    props = torch.cuda.get_device_properties(dev)
    dev_id = str(getattr(props, "uuid", "") or "") or \
        f"pci {getattr(props, 'pci_domain_id', 0)}:{getattr(props, 'pci_bus_id', '?')}:{getattr(props, 'pci_device_id', '?')}"
    dev_ids = [dev_id]
    if use_dist and world > 1:
        dev_ids = [None] * world
        dist.all_gather_object(dev_ids, dev_id)
    nccl_version = None
    if use_dist and args.backend == "nccl":
        try:
            nccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:   # noqa: BLE001
            nccl_version = "unknown"
    # ---- roofline of the dominant kernel, per launch, this rank's share ----
    my_pairs = n_pairs / world
    # N = 1: a step IS one launch of the kernel, and `frac` is priced on the interval `value` is computed from (the host's
    # clock around the K steps between the two fences: ms_per_step); the HIP-event span of the same K launches -- a few
    # per cent shorter: it excludes the replay's launch and the closing fence -- is printed beside it as
    # frac_kernel_span.  N > 1: a step also holds the exchange, so the kernel's own events price it.
    span_s = kern_ms_max * 1e-3
    kern_s = (dt / args.steps) if not use_dist else span_s
    alg_bytes = float(cell_bytes) * my_pairs + lib.ldx_plane_bytes(n_snps, n_hap)   # result cells + the ALT plane read once
    alg_ops = 2.0 * n_hap * my_pairs                                    # multiply-adds x 2 (2*H per pair)
    lane_ops = 2.0 * math.ceil(n_hap / 32) * my_pairs                   # v_and_b32 + v_bcnt_u32_b32 per 32 haplotypes
    traffic, traffic_src, traffic_head, traffic_ksrc = None, None, None, None
    from ld_tools_amd.build import source_digest
    kernel_src = source_digest()            # sha256 of csrc/ + include/ldx.h as they are in this tree
    tfile = ROOT / "profiles" / "traffic.json"
    if tfile.exists():
        try:
            rec = json.loads(tfile.read_text())
            if (rec.get("workload") == f"ld_triangle {n_snps}x{n_hap}" and rec.get("gpus") == world
                    and rec.get("path") == path and rec.get("fmt") == fmt):
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_src = f"profiles/traffic.json ({rec.get('profile', '?')}; rocprofv3 PMC, not measured in this run)"
                traffic_head = rec.get("head")               # the commit the counters were taken at
                traffic_ksrc = rec.get("kernel_src")         # ... and the digest of the kernel sources then
        except (ValueError, OSError):
            pass
    provenance = {"traffic_head": traffic_head, "traffic_kernel_src": traffic_ksrc, "kernel_src": kernel_src,
                  "traffic_is_of_this_kernel": (traffic_ksrc == kernel_src) if traffic is not None else None}
    hbm = {"bound": "hbm", "achieved": alg_bytes / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": alg_bytes / kern_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
           "algorithmic_bytes": alg_bytes, **provenance}
    if path in ("fp4", "mfma"):   # the counting runs on the matrix pipe: that ceiling governs
        peak = MFMA_FP4_PEAK_TOPS if path == "fp4" else MFMA_I8_PEAK_TOPS
        roofline = {"bound": "mfma", "achieved": alg_ops / kern_s / 1e12, "peak": peak, "unit": "TOP/s",
                    "frac": alg_ops / kern_s / 1e12 / peak, "traffic": traffic, "traffic_source": traffic_src, **provenance,
                    "kernel": "triangle_mfma_kernel", "kernel_ms": kern_s * 1e3, "kernel_ms_min_rank": kern_ms_min,
                    "kernel_ms_event_span": kern_ms_max, "frac_kernel_span": alg_ops / span_s / 1e12 / peak,
                    "interval": ("ms_per_step (host clock around the K graph-replayed launches, the interval of `value`)"
                                 if not use_dist else "HIP events around each launch on its stream"),
                    "algorithmic_ops": alg_ops, "ops_per_pair": 2 * n_hap,
                    "pipe": ("v_mfma_f32_32x32x64_f8f6f4, FP4 operands (dense peak 10 POP/s)" if path == "fp4"
                             else "v_mfma_i32_32x32x32_i8 (dense peak 5 POP/s)"),
                    "frac_of_int8_peak": alg_ops / kern_s / 1e12 / MFMA_I8_PEAK_TOPS,
                    "note": "multiply-adds counted as 2 ops; 0/1 operands, exact integer result"}
    else:
        roofline = dict(hbm, kernel="triangle_kernel", kernel_ms=kern_ms_max)
    line = {
        "metric": "SNP-pairs/sec (r2+D')",
        "value": value,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,               # the MEDIAN of the R timed regions below (value, roofline: from it)
        "ms_per_step_min": dt_min / args.steps * 1e3,
        "ms_per_step_runs": [x / args.steps * 1e3 for x in per_region],   # every timed region of K steps, in run order
        "timing": f"{repeats} timed regions of {args.steps} steps each (barrier + synchronize on both sides, max over ranks "
                  "per region); ms_per_step / value / roofline.frac from the median region",
        "cold_ms_per_step": cold_dt / args.steps * 1e3,     # the same K steps from idle clocks, eager launches
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": {"fp4": "fp4 (E2M1 0/1 operands, exact f32 accumulation)", "mfma": "int8", "popcount": "u32"}[path],
        "data": "synthetic",
        "config": {"workload": f"ld_triangle {n_snps}x{n_hap}", "n_snps": n_snps, "n_hap": n_hap,
                   "pairs_per_step": n_pairs,
                   "output": ("4 B/pair in HBM: uint16 k = round(x, 4) * 10^4 for r2 and D' (lossless)" if fmt == "k16"
                              else "8 B/pair in HBM: float32 r2, float32 D', rounded to 4 decimals"),
                   "kernel_path": {"fp4": "FP4 MFMA counts + fp32 / fp64 / mirror epilogue tiers",
                                   "mfma": "int8 MFMA counts + fp64 epilogue",
                                   "popcount": "AND+popcount counts + fp64 epilogue"}[path],
                   "launch": (("HIP graph of the K steps" if graph is not None else "eager launches") +
                              f"; every cell of the timed region's output verified against the {check_path} kernel"),
                   "settle_steps": settle_done,   # untimed, beyond --warmup: the timed region starts at sustained clocks
                   "sharding": "none" if world == 1 else f"row-block shards, all-gather, pass list / {world}",
                   # ranks that met over RCCL (null when the group is not an RCCL group: gloo rehearsals, no group at N = 1)
                   "rccl_ranks": dist.get_world_size() if (use_dist and args.backend == "nccl") else None,
                   "group_ranks": dist.get_world_size() if use_dist else 1,
                   "backend": args.backend if use_dist else None,
                   "nccl_version": nccl_version,
                   "n_distinct_devices": len(set(dev_ids)),
                   "devices": sorted(set(str(d) for d in dev_ids)),
                   "exchange": ("none" if not use_dist else "per step, overlapped with the previous step's kernel"
                                if pipe is not None else "per step, before the kernel")},
        "roofline": roofline,
        "roofline_hbm": hbm,      # the metric's "% HBM roofline": result bytes + one read of the packed plane
    }
    if path == "popcount":   # the popcount path is bound by the integer VALU, not by HBM
        line["roofline_valu"] = {"bound": "valu-int", "achieved": lane_ops / kern_s / 1e12, "peak": VALU_PEAK_TLANEOPS,
                                 "unit": "T lane-ops/s", "frac": lane_ops / kern_s / 1e12 / VALU_PEAK_TLANEOPS,
                                 "ops_per_pair": 2 * math.ceil(n_hap / 32)}
    if graph is not None and use_dist:   # the graph's span is all there is: the exchange rides in the per-step figure
        roofline["kernel_ms_includes_exchange"] = True
    elif use_dist:                       # eager: exchange = step time minus the kernel's own events
        line["config"]["exchange_ms_per_step"] = max(0.0, dt / args.steps * 1e3 - kern_ms_max)
    if use_dist:
        # Per rank: its kernel time, what its step spends outside the kernel (the exposed part of the
        # exchange) and its share of the pairs -- the first record of a real multi-GPU node then shows at a glance whether
        # the contiguous unit ranges are balanced and whether the all-gather hides under the kernel.
        from ld_tools_amd._lib import UNIT_PAIRS as _UP
        mine = {"rank": rank, "kernel_ms": kern_ms, "step_ms": runs[mid][0] / args.steps * 1e3,
                "exchange_ms_per_step": (None if graph is not None else max(0.0, runs[mid][0] / args.steps * 1e3 - kern_ms)),
                "units": int(u1 - u0), "cells": int(u1 - u0) * _UP,
                "pairs": ldist.pairs_in_units(n_snps, u0, u1), "device": dev_id}
        per_rank = [None] * world
        if world > 1:
            dist.all_gather_object(per_rank, mine)
        else:
            per_rank = [mine]
        line["per_rank"] = per_rank

    def leg(leg_path, leg_fmt, reps):
        """Another kernel path / cell format on the same resident panel (N = 1), eager launches at settled clocks."""
        res = ld_triangle(panel, fmt=leg_fmt, path=leg_path)
        for _ in range(max(3, reps // 4)):
            ld_triangle(panel, out=res, fmt=leg_fmt, path=leg_path)
        torch.cuda.synchronize()
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            ld_triangle(panel, out=res, fmt=leg_fmt, path=leg_path)
        c.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(c) / reps
        o = {"ms": ms, "pairs_per_s": n_pairs / (ms * 1e-3), "fmt": leg_fmt}
        if leg_fmt == fmt:      # the same cells as the headline's, from another kernel: compared, every one of them
            o["results_equal"] = bool(torch.equal(res.cells.view(torch.int32), out.cells.view(torch.int32)))
            o["verified_against"] = "the timed region's output, every cell"
        if leg_path == "popcount":
            o["frac_of_valu_int_peak"] = lane_ops / (ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS
        else:
            o["frac_of_pipe_peak"] = alg_ops / (ms * 1e-3) / 1e12 / (MFMA_FP4_PEAK_TOPS if leg_path == "fp4" else MFMA_I8_PEAK_TOPS)
        return o

    if world == 1 and not use_dist and not args.no_extra_legs:
        # north_star: "MFMA only if ... proves faster than the popcount path" -- all three on the driver-run record
        reps = max(10, min(args.steps, 100))
        other = "ld32" if fmt == "k16" else "k16"
        line["other_paths"] = {f"fp4_{other}": leg("fp4", other, reps),
                               "mfma_int8": leg("mfma", fmt, reps),
                               "popcount": leg("popcount", fmt, max(5, reps // 4))}
        # the headline's cells against an INDEPENDENT kernel on this very box (AND + popcount counts, fp64 epilogue: it
        # shares neither the counting nor the fp32 tier with the FP4 kernel); a difference ends the run with RC_VERIFY below
        if line["other_paths"]["mfma_int8"].get("results_equal"):
            line["config"]["launch"] += " (and the int8 MFMA kernel's timed leg)"
        # Independent batches on two streams (two result buffers): what a driver that walks chromosomes or tables gets (the
        # shells' table workers each own a stream).  The workgroups of batch k + 1 start while those of batch k drain their
        # last passes.  NOT the headline: `value` and `roofline` are one launch after the other on one stream.  Since
        # round 5 under the headline's own conditions (the leg used to run 20-100 eager-captured batches
        # right behind the popcount leg, without settling, and read 14 % SLOWER than the headline while the same-box
        # comparison of tools/gpu_streams.py reads 10 % faster): >= 200 batches per graph, the same settling load, and the
        # one-stream graph timed beside it in the same loop, interleaved, medians of three.
        try:
            import statistics

            nb = max(200, args.steps)
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            outs = [ld_triangle(panel, fmt=fmt), ld_triangle(panel, fmt=fmt)]

            def two_streams(count):
                cur = torch.cuda.current_stream()      # inside a capture: the capturing stream (fork / join around it)
                for st in streams:
                    st.wait_stream(cur)
                for k in range(count):
                    with torch.cuda.stream(streams[k & 1]):
                        ld_triangle(panel, out=outs[k & 1], fmt=fmt)
                for st in streams:
                    cur.wait_stream(st)

            def one_stream(count):
                for _ in range(count):
                    ld_triangle(panel, out=out, fmt=fmt)

            two_streams(4)                              # every stream has launched before it is captured
            torch.cuda.synchronize()
            runs, how = {"one": (lambda: one_stream(nb)), "two": (lambda: two_streams(nb))}, "eager launches"
            if graph is not None:           # like the headline: the launches as one HIP graph (fork / join across the streams)
                try:
                    gs = {}
                    for name, fn in (("one", one_stream), ("two", two_streams)):
                        g2 = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g2):
                            fn(nb)
                        g2.replay()
                        torch.cuda.synchronize()
                        gs[name] = g2
                    runs, how = {k: v.replay for k, v in gs.items()}, "one HIP graph each"
                except Exception:           # noqa: BLE001
                    torch.cuda.synchronize()
            for _ in range(max(1, int(settle_done / nb))):      # the headline's settling load
                runs["one"]()
            torch.cuda.synchronize()
            for o in outs:
                o.cells.fill_(-1)
            got = {"one": [], "two": []}
            for _ in range(3):
                for name in ("one", "two"):
                    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    runs[name]()
                    c.record()
                    torch.cuda.synchronize()
                    got[name].append(a.elapsed_time(c) / nb)
            ms1, ms2 = statistics.median(got["one"]), statistics.median(got["two"])
            same = all(torch.equal(o.cells.view(torch.int32), out.cells.view(torch.int32)) for o in outs)
            line["other_paths"]["two_streams"] = {
                "ms_per_batch": ms2, "pairs_per_s": n_pairs / (ms2 * 1e-3), "fmt": fmt, "results_equal": bool(same),
                "one_stream_same_conditions_ms": ms1, "batches_per_graph": nb, "rounds": 3,
                "note": f"independent batches alternating on two HIP streams against the same batches on one stream: {how}, "
                        "interleaved after the headline's settling load, medians"}
            del outs
        except Exception as exc:   # noqa: BLE001  (an extra)
            line["other_paths"]["two_streams"] = {"error": f"{type(exc).__name__}: {exc}"}
    if world == 1 and not use_dist and not args.no_other_workloads:
        line["other_workloads"] = other_workloads(torch, dev, fmt, codes_local, args.other_scale, args.debug_corrupt_other)
    # every extra leg that compares two results is held to the headline's rule: a mismatch is a wrong RESULT -- no
    # throughput line, exit code RC_VERIFY
    bad = [k for grp in ("other_paths", "other_workloads") for k, v in line.get(grp, {}).items()
           if isinstance(v, dict) and v.get("results_equal") is False]
    if bad:
        print(f"bench.py: {bad} did not reproduce the independently computed result", file=sys.stderr, flush=True)
        print(f"{VERIFY_MARK} rank {rank} {bad}", flush=True)
        os._exit(RC_VERIFY)
    if world > 1 and not args.no_single_gpu_leg:
        # strong-scaling reference: the WHOLE workload on rank 0's GPU alone, a few steps (the other ranks wait)
        single = None
        if rank == 0:
            try:
                res1 = ld_triangle(panel, fmt=fmt)
                torch.cuda.synchronize()
                a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    ld_triangle(panel, out=res1, fmt=fmt)
                c.record()
                torch.cuda.synchronize()
                ms1 = a.elapsed_time(c) / 3
                single = {"ms_per_step": ms1, "pairs_per_s": n_pairs / (ms1 * 1e-3)}
                del res1
            except Exception as exc:   # noqa: BLE001  (an extra: e.g. not enough memory for the whole triangle)
                single = {"error": f"{type(exc).__name__}: {exc}"}
        dist.barrier()
        if rank == 0:
            line["config"]["single_gpu_same_workload"] = single
            if single and "pairs_per_s" in single:
                line["config"]["speedup_over_one_gpu"] = value / single["pairs_per_s"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        need = max(args.cpu_sample_snps, 1536)
        host = codes_local[:need].cpu().numpy()
        line["cpu_baseline"] = cpu_baseline(host, args.cpu_sample_snps)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
