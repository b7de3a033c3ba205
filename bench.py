#!/usr/bin/env python3
"""Benchmark of the pairwise-LD hot path on MI355X: SNP-pairs/s producing r^2 + D'.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--snps S] [--haps H]

One "step" = one full pass of the hot path over one synthetic panel that is already resident in HBM
as packed slab shards: (N > 1: all-gather of the shards over RCCL) -> ld_triangle kernel over this
rank's share of the pass list -> 8 bytes per pair (float32 r^2, float32 D', 4-decimal) written to HBM.

Workload (BASELINE.json): N = 1 -> configs[1], ld_triangle 10 000 SNPs x 5008 haplotypes.  N > 1 keeps
the pairs per GPU constant (weak scaling): S = 10 000 * sqrt(N) SNPs rounded up to whole 128-row slabs,
row-block shards packed per rank, exchanged by all-gather, unit list split evenly (ld_tools_amd/dist.py).
--snps overrides S (e.g. --snps 100000 for configs[3]).

Timing: W warm-up steps, then --settle-steps more untimed steps (the shader clock needs tens of ms of load to
settle; reported as config.settle_steps), then EXACTLY K steps between barrier + synchronize on both sides, max
over ranks.  The K steps are replayed as one HIP graph (eager with --no-graph); the output is NaN-filled before the
timed region and compared bit for bit with a separately computed result after it.

Prints ONE JSON line on rank 0 (see the driver contract): whole-job pairs/s, ms per step, the roofline
object of the dominant kernel (HIP events on the launch stream, live) and, at N = 1, the CPU baseline
(the pure-Python restatement of the reference's list/zip/count algorithm, 1 core, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_I8_PEAK_TOPS = 5000.0                # ... dense I8 MFMA = 2x the BF16 rate per clock = ~5 POP/s (no sparsity)
VALU_PEAK_TLANEOPS = 256 * 64 * 2.4e9 / 1e12   # AND / BCNT have no packed form: 64 lanes/clk/CU at 2.4 GHz = 39.3 T


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--snps", type=int, default=0, help="panel size (default: 10000 * sqrt(gpus))")
    ap.add_argument("--haps", type=int, default=5008)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--force-dist", action="store_true", help="run the sharded path (process group + all-gather) even at world 1")
    ap.add_argument("--cpu-sample-snps", type=int, default=320)
    ap.add_argument("--no-graph", action="store_true", help="launch the steps eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: finish each step's all-gather before its kernel instead of overlapping it with the "
                         "previous step's kernel")
    ap.add_argument("--overlap", action="store_true", help="use the overlapped exchange even in a one-rank group (rehearsals)")
    ap.add_argument("--settle-steps", type=int, default=600,
                    help="untimed steps run right before the timed region, on top of --warmup, so that it starts at "
                         "sustained clocks (the shader clock needs tens of ms of load to settle; 0 = off)")
    ap.add_argument("--path", default="auto", choices=("auto", "fp4", "mfma", "popcount"),
                    help="kernel behind ld_triangle (auto = the int8 MFMA kernel; results are identical)")
    return ap.parse_args()


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


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
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
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from ld_tools_amd import PackedPanel, dist as ldist, ld_triangle, ops, synth
    from ld_tools_amd._lib import lib

    ops.set_triangle_path(args.path)
    mfma = args.path != "popcount"
    n_hap = args.haps
    if args.snps:
        n_snps = args.snps
    elif world == 1:
        n_snps = 10000                                                   # BASELINE.json configs[1]
    else:
        n_snps = int(math.ceil(10000 * math.sqrt(world) / 128.0)) * 128  # constant pairs per GPU
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
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

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
            ev0[k].record()
        out = ld_triangle(panel, unit_range=(u0, u1), out=out)
        if k is not None:
            ev1[k].record()

    def run_steps(count, timed=False):
        for k in range(count):
            step(k if timed else None, first=(k == 0), last=(k == count - 1))

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    fence()
    # The K steps are K back-to-back launches of a 0.25 ms kernel (plus, for N > 1, the exchange's all-gather and a
    # handful of small copies): a launch-bound inner loop, captured once into a HIP graph and replayed inside the
    # timed region.  The kernel's duration is then (graph span - exchange share) / K from HIP events on the replay
    # stream at N = 1; any failure to capture falls back to eager launches.
    graph = None
    if not args.no_graph and (not use_dist or args.backend == "nccl"):   # RCCL collectives capture; gloo ones do not
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
    # Clock settling (untimed, reported in config): a 20-step timed region is 5 ms, far shorter than the tens of ms the
    # shader clock takes to settle under load, and would read 15 % slow whatever the launch method.
    # A fixed step count, not a time, so that every rank issues the same collectives.
    settle_done = 0
    if args.settle_steps > 0:
        if graph is not None:
            for _ in range((args.settle_steps + args.steps - 1) // args.steps):
                graph.replay()
                settle_done += args.steps
        else:
            run_steps(args.settle_steps)
            settle_done = args.settle_steps
    if out is not None:
        out.ld32.fill_(float("nan"))       # the timed steps must produce every result again (checked below)
    fence()
    t0 = time.perf_counter()
    if graph is not None:
        ev0[0].record()
        graph.replay()
        ev1[0].record()
    else:
        run_steps(args.steps, timed=True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if graph is not None:
        kern_ms = ev0[0].elapsed_time(ev1[0]) / args.steps
    else:
        kern_ms = sum(a.elapsed_time(b_) for a, b_ in zip(ev0, ev1)) / args.steps
    if world > 1:
        t = torch.tensor([kern_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kern_ms = float(t.item())

    # the output of the timed region against a separately computed result: nothing was skipped or left stale
    check = ld_triangle(panel, unit_range=(u0, u1))
    torch.cuda.synchronize()
    if not torch.equal(check.ld32.view(torch.int32), out.ld32.view(torch.int32)):
        raise SystemExit("bench.py: the timed steps did not reproduce the triangle (stale or skipped work)")
    del check

    value = n_pairs * args.steps / dt
    # ---- roofline of the dominant kernel, per launch, this rank's share ----
    my_pairs = n_pairs / world
    kern_s = kern_ms * 1e-3
    alg_bytes = 8.0 * my_pairs + lib.ldx_plane_bytes(n_snps, n_hap)     # 8 B/pair out + the ALT plane read once
    alg_ops = 2.0 * n_hap * my_pairs                                    # int8 multiply-adds x 2 (2*H per pair)
    lane_ops = 2.0 * math.ceil(n_hap / 32) * my_pairs                   # v_and_b32 + v_bcnt_u32_b32 per 32 haplotypes
    traffic = None
    tfile = ROOT / "profiles" / "traffic.json"
    if tfile.exists():
        try:
            rec = json.loads(tfile.read_text())
            if (rec.get("workload") == f"ld_triangle {n_snps}x{n_hap}" and rec.get("gpus") == world
                    and rec.get("path", "mfma") == ("mfma" if mfma else "popcount")):
                traffic = rec.get("hbm_bytes_per_launch")
        except (ValueError, OSError):
            pass
    hbm = {"bound": "hbm", "achieved": alg_bytes / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": alg_bytes / kern_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": alg_bytes}
    if mfma:   # the counting runs on the matrix pipe: that ceiling governs (5 POP/s / 10 016 ops = 5.0e11 pairs/s)
        roofline = {"bound": "mfma", "achieved": alg_ops / kern_s / 1e12, "peak": MFMA_I8_PEAK_TOPS, "unit": "TFLOP/s",
                    "frac": alg_ops / kern_s / 1e12 / MFMA_I8_PEAK_TOPS, "traffic": traffic,
                    "kernel": "triangle_mfma_kernel", "kernel_ms": kern_ms, "algorithmic_ops": alg_ops,
                    "ops_per_pair": 2 * n_hap, "note": "int8 multiply-adds counted as 2 ops (integer, not floating point)"}
    else:
        roofline = dict(hbm, kernel="triangle_kernel", kernel_ms=kern_ms)
    line = {
        "metric": "SNP-pairs/sec (r2+D')",
        "value": value,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int8" if mfma else "u32",
        "data": "synthetic",
        "config": {"workload": f"ld_triangle {n_snps}x{n_hap}", "n_snps": n_snps, "n_hap": n_hap,
                   "pairs_per_step": n_pairs, "output": "8 B/pair (f32 r2, f32 D', rounded to 4 decimals) in HBM",
                   "kernel_path": "int8 MFMA counts + f64 epilogue" if mfma else "AND+popcount counts + f64 epilogue",
                   "launch": "HIP graph of the K steps, output verified after the timed region" if graph is not None else "eager",
                   "settle_steps": settle_done,   # untimed, beyond --warmup: the timed region starts at sustained clocks
                   "sharding": "none" if world == 1 else f"row-block shards, all-gather, pass list / {world}",
                   "exchange": ("none" if not use_dist else "per step, overlapped with the previous step's kernel"
                                if pipe is not None else "per step, before the kernel")},
        "roofline": roofline,
        "roofline_hbm": hbm,      # the metric's "% HBM roofline": output bytes + one read of the packed plane
    }
    if not mfma:   # the popcount path is bound by the integer VALU, not by HBM
        line["roofline_valu"] = {"bound": "valu-int", "achieved": lane_ops / kern_s / 1e12, "peak": VALU_PEAK_TLANEOPS,
                                 "unit": "T lane-ops/s", "frac": lane_ops / kern_s / 1e12 / VALU_PEAK_TLANEOPS,
                                 "ops_per_pair": 2 * math.ceil(n_hap / 32)}
    if graph is not None and use_dist:   # the graph's span is all there is: the exchange rides in the per-step figure
        roofline["kernel_ms_includes_exchange"] = True
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        need = max(args.cpu_sample_snps, 1536)
        host = codes_local[:need].cpu().numpy()
        line["cpu_baseline"] = cpu_baseline(host, args.cpu_sample_snps)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
