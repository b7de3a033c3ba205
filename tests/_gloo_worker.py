"""Worker for tests/test_dist_gloo.py: rehearsal of the sharded triangle on CPU (gloo), world size 2 or 8.

Each rank "packs" only its slab shard of a synthetic panel (numpy, in the tiled byte layout of
include/ldx.h), the shards are all-gathered with ld_tools_amd.dist.gather_shards, every rank checks
that the gathered plane equals the plane packed from the whole panel, then computes ITS unit range of
the triangle with the C oracle and rank 0 checks that the union of the ranks' cells is the full
triangle, each cell exactly once and equal to the single-process result.
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from ld_tools_amd import dist as ldist  # noqa: E402
from ld_tools_amd import synth  # noqa: E402
from oracle import c_oracle  # noqa: E402


def tiled_plane(codes: np.ndarray, n_hap: int) -> np.ndarray:
    """uint8 tiled ALT plane of include/ldx.h for int8 codes [rows][n_hap] (rows padded to slabs)."""
    rows = codes.shape[0]
    slabs, chunks = (rows + 127) // 128, ldist.n_chunks(n_hap)
    bits = np.zeros((slabs * 128, chunks * 128), dtype=np.uint8)
    bits[:rows, :n_hap] = codes == 1
    by = np.packbits(bits, axis=1, bitorder="little").reshape(slabs, 128, chunks, 16)
    return np.ascontiguousarray(by.transpose(0, 2, 1, 3)).reshape(-1)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    for n_snps, n_hap in [(512, 200), (700, 137), (600, 70)]:       # equal slab shards, ragged last slab, uneven slab counts
        full_codes = synth.synth_codes_host(n_snps, n_hap, seed=3, miss=0.01)
        parts = ldist.slab_partition(n_snps, world)
        b, e = parts[rank]
        mine = synth.synth_codes_host(e - b, n_hap, seed=3, miss=0.01, snp_offset=b)   # rank-local ingest
        assert np.array_equal(mine, full_codes[b:e])
        shard = torch.from_numpy(tiled_plane(mine, n_hap))
        slab_bytes = ldist.n_chunks(n_hap) * 128 * 16
        sizes = [((pe - pb + 127) // 128) * slab_bytes for (pb, pe) in parts]
        dst = torch.zeros(sum(sizes), dtype=torch.uint8)
        ldist.gather_shards(dst, shard, sizes)
        want = tiled_plane(full_codes, n_hap)
        assert np.array_equal(dst.numpy(), want), "gathered plane differs from the whole-panel plane"

        # the one-collective exchange of bench.py / all_gather_panel (fused_gather), with and without the REF plane,
        # twice through the same staging buffers
        slabs = [(pe - pb + 127) // 128 for (pb, pe) in parts]
        npad_local, npad_full = slabs[rank] * 128, sum(slabs) * 128
        acnt = np.zeros(npad_local, dtype=np.int32)
        rcnt = np.zeros(npad_local, dtype=np.int32)
        acnt[: e - b] = (mine == 1).sum(axis=1)
        rcnt[: e - b] = (mine == 0).sum(axis=1)
        ref_shard = torch.from_numpy(tiled_plane((mine == 0).astype(np.int8), n_hap))
        for with_ref in (False, True):
            local = {"alt": shard, "acnt": torch.from_numpy(acnt), "rcnt": torch.from_numpy(rcnt)}
            full = {"alt": torch.zeros(sum(sizes), dtype=torch.uint8), "acnt": torch.zeros(npad_full, dtype=torch.int32),
                    "rcnt": torch.zeros(npad_full, dtype=torch.int32)}
            if with_ref:
                local["ref"] = ref_shard
                full["ref"] = torch.zeros(sum(sizes), dtype=torch.uint8)
            stage = None
            for _ in range(2):
                stage = ldist.fused_gather(full, local, slabs, slab_bytes, stage=stage)
            for piece in full.values():                # and once more through the asynchronous halves (PanelPipeline's calls)
                piece.zero_()
            stage, work = ldist.fused_gather_start(local, slabs, slab_bytes, torch.device("cpu"), None, stage, with_ref,
                                                   async_op=True)
            assert work is not None
            ldist.fused_gather_finish(full, stage, slabs, slab_bytes, work)
            assert np.array_equal(full["alt"].numpy(), want), "fused gather: ALT plane"
            fa = np.zeros(npad_full, dtype=np.int32)
            fr_ = np.zeros(npad_full, dtype=np.int32)
            fa[:n_snps] = (full_codes == 1).sum(axis=1)
            fr_[:n_snps] = (full_codes == 0).sum(axis=1)
            assert np.array_equal(full["acnt"].numpy(), fa) and np.array_equal(full["rcnt"].numpy(), fr_), "fused gather: counts"
            if with_ref:
                assert np.array_equal(full["ref"].numpy(), tiled_plane((full_codes == 0).astype(np.int8), n_hap))

        # sharded triangle: this rank's unit range, cells through the oracle
        u0, u1 = ldist.unit_partition(n_snps, world)[rank]
        rows, cols = ldist.unit_cells(n_snps, u0, u1)
        p = c_oracle.Panel(full_codes)
        n11 = np.array([int(p.pair_counts(r, r + 1, c, c + 1)[0, 0]) for r, c in zip(rows[::97], cols[::97])])
        cover = torch.zeros((n_snps, n_snps), dtype=torch.int32)
        cover[torch.from_numpy(rows), torch.from_numpy(cols)] = 1
        dist.all_reduce(cover)
        if rank == 0:
            assert torch.equal(cover, torch.tril(torch.ones_like(cover), -1)), "cells not covered exactly once"
        t = p.triangle(want=("n11",))["n11"]
        assert np.array_equal(n11, t[rows[::97], cols[::97]])
        pairs = torch.tensor([len(rows)], dtype=torch.int64)
        dist.all_reduce(pairs)
        assert int(pairs) == n_snps * (n_snps - 1) // 2
    # configs[3]'s geometry with a real process group (no group larger than 2 had ever run dist.py): 100 000
    # SNPs -> 782 slabs; over 8 ranks 98, ..., 98, 96 -- the uneven branch of fused_gather_finish -- through the exchange
    # bench.py uses (PanelPipeline's fused_gather_start / fused_gather_finish, asynchronous), twice through the same staging
    # buffers; 16 haplotypes keep the plane small (the slab geometry depends on the SNP count only).  Then the unit ranges
    # of the ranks: contiguous, in rank order, equal within one unit, covering every unit of the triangle once.
    n_snps, n_hap = 100000, 16
    parts = ldist.slab_partition(n_snps, world)
    slabs = [(pe - pb + 127) // 128 for (pb, pe) in parts]
    assert sum(slabs) == 782 and (world != 8 or slabs == [98] * 7 + [96])
    b, e = parts[rank]
    mine = synth.synth_codes_host(e - b, n_hap, seed=5, snp_offset=b) if e > b else np.zeros((0, n_hap), np.int8)
    slab_bytes = ldist.n_chunks(n_hap) * 128 * 16
    acnt = np.zeros(slabs[rank] * 128, dtype=np.int32)
    rcnt = np.zeros(slabs[rank] * 128, dtype=np.int32)
    acnt[: e - b] = (mine == 1).sum(axis=1)
    rcnt[: e - b] = (mine == 0).sum(axis=1)
    local = None if e == b else {"alt": torch.from_numpy(tiled_plane(mine, n_hap)), "acnt": torch.from_numpy(acnt),
                                 "rcnt": torch.from_numpy(rcnt)}
    full = {"alt": torch.zeros(782 * slab_bytes, dtype=torch.uint8), "acnt": torch.zeros(782 * 128, dtype=torch.int32),
            "rcnt": torch.zeros(782 * 128, dtype=torch.int32)}
    stage = None
    for _ in range(2):
        for piece in full.values():
            piece.zero_()
        stage, work = ldist.fused_gather_start(local, slabs, slab_bytes, torch.device("cpu"), None, stage, False, async_op=True)
        ldist.fused_gather_finish(full, stage, slabs, slab_bytes, work)
    whole = synth.synth_codes_host(n_snps, n_hap, seed=5)
    assert np.array_equal(full["alt"].numpy(), tiled_plane(whole, n_hap)), "configs[3] geometry: gathered ALT plane"
    fa = np.zeros(782 * 128, dtype=np.int32)
    fa[:n_snps] = (whole == 1).sum(axis=1)
    assert np.array_equal(full["acnt"].numpy(), fa), "configs[3] geometry: gathered counts"
    units = ldist.unit_partition(n_snps, world)
    total = ldist.triangle_units(n_snps)
    mine_u = torch.tensor([units[rank][0], units[rank][1]], dtype=torch.int64)
    all_u = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(all_u, mine_u)
    edges = [int(x) for t in all_u for x in t]
    assert edges[0] == 0 and edges[-1] == total and all(edges[2 * r + 1] == edges[2 * r + 2] for r in range(world - 1))
    sizes_u = [edges[2 * r + 1] - edges[2 * r] for r in range(world)]
    assert max(sizes_u) - min(sizes_u) <= 1
    # ld_area hit lists: variable-length shards (some of them empty) gathered in rank order
    for pattern in ([5, 0], [3, 11], [0, 0], [7, 7]):
        sizes = [pattern[r % 2] + (r // 2) * (1 if pattern[r % 2] else 0) for r in range(world)]
        k = sizes[rank]
        base = sum(sizes[:rank])
        q = torch.arange(base, base + k, dtype=torch.int64)
        o = q * 3 + 1
        ld = torch.stack([q.to(torch.float32) * 1e-4, -torch.zeros(k)], dim=1)     # -0.0 = the reference's int 0
        gq, go, gl = ldist.gather_hits(q, o, ld)
        tot = sum(sizes)
        assert torch.equal(gq, torch.arange(tot, dtype=torch.int64)) and torch.equal(go, gq * 3 + 1)
        assert gl.shape == (tot, 2) and torch.equal(gl[:, 0], gq.to(torch.float32) * 1e-4)
        assert bool(torch.signbit(gl[:, 1]).all()) and bool((gl[:, 1] == 0).all())
    dist.barrier()
    if rank == 0:
        print("GLOO_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
