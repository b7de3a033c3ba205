"""Worker for tests/test_gpu_dist.py: world_size-2 rehearsal of the sharded ld_triangle on ONE card.

Both ranks use cuda:0 and the gloo backend (RCCL needs one device per rank; the exchange code path is the same
collective call).  Each rank packs only its slab shard, the shards are exchanged with all_gather_panel (one fused
all-gather), every rank runs ld_triangle on its unit range of the gathered panel, and compares planes, counts and
its shard of the triangle with the panel packed from all rows in this process.
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from ld_tools_amd import PackedPanel, ld_triangle, synth  # noqa: E402
from ld_tools_amd import dist as ldist  # noqa: E402
from ld_tools_amd._lib import UNIT_PAIRS  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    for n_snps, n_hap in [(1024, 5008), (1500, 1008), (700, 333)]:      # equal shards, uneven slab counts, ragged last slab
        b, e = ldist.slab_partition(n_snps, world)[rank]
        mine = synth.synth_codes_device(e - b, n_hap, seed=5, miss=0.002, snp_offset=b, device=dev)
        local = PackedPanel.from_codes(mine)
        panel = None
        for _ in range(2):                                              # second call re-uses panel and staging buffers
            panel = ldist.all_gather_panel(local, n_snps, n_hap, out=panel)
        whole = PackedPanel.from_codes(synth.synth_codes_device(n_snps, n_hap, seed=5, miss=0.002, device=dev))
        assert torch.equal(panel.alt, whole.alt), "gathered ALT plane"
        assert torch.equal(panel.acnt, whole.acnt) and torch.equal(panel.rcnt, whole.rcnt), "gathered counts"
        assert torch.equal(panel.fa, whole.fa) and torch.equal(panel.q, whole.q), "frequency vectors"
        u0, u1 = ldist.unit_partition(n_snps, world)[rank]
        part = ld_triangle(panel, unit_range=(u0, u1), want_n11=True)
        full = ld_triangle(whole, want_n11=True)
        assert torch.equal(part.ld32.view(torch.int32), full.ld32[u0 * UNIT_PAIRS: u1 * UNIT_PAIRS].view(torch.int32)), "shard results"
        assert torch.equal(part.n11, full.n11[u0 * UNIT_PAIRS: u1 * UNIT_PAIRS]), "shard counts"
        with_ref = ldist.all_gather_panel(local, n_snps, n_hap, with_ref=True)
        assert torch.equal(with_ref.ref, whole.ref), "gathered REF plane"
        # the double-buffered pipeline: exchange of batch k + 1 in flight while batch k's kernel runs
        pipe = ldist.PanelPipeline(n_snps, n_hap, dev)
        pipe.start(local)
        for k in range(4):
            got = pipe.finish()
            if k < 3:
                pipe.start(local)
            assert got is pipe.panels[k % 2]
            r = ld_triangle(got, unit_range=(u0, u1))
            assert torch.equal(got.alt, whole.alt) and torch.equal(got.acnt, whole.acnt) and torch.equal(got.q, whole.q)
            assert torch.equal(r.ld32.view(torch.int32), part.ld32.view(torch.int32)), "pipelined shard results"
        try:
            pipe.finish()
            raise AssertionError("finish without an exchange in flight must fail")
        except RuntimeError:
            pass
    # ld_area sharded by query: the gathered hit list is the single-process hit list, on every rank
    from ld_tools_amd import ld_area
    n_snps, n_hap = 3000, 1008
    whole = PackedPanel.from_codes(synth.synth_codes_device(n_snps, n_hap, seed=8, miss=0.002, device=dev))
    rng = np.random.RandomState(3)
    pos = np.cumsum(rng.choice([0, 1, 40, 2500], size=n_snps, p=[0.05, 0.35, 0.4, 0.2])) + 1
    for queries, flank, measure, thres in [(None, 20000, "r_square", 0.8), (list(range(0, n_snps, 5)), 100000, "d_prime", 1.0),
                                           ([17], 500, "r_square", 0.0), (None, 0, "r_square", 0.0)]:
        want = ld_area(whole, pos, queries, flank, measure, thres)
        got = ldist.ld_area_sharded(whole, pos, queries, flank, measure, thres)
        assert torch.equal(got.query, want.query) and torch.equal(got.oppos, want.oppos), "sharded ld_area rows"
        assert torch.equal(got.ld32.view(torch.int32), want.ld32.view(torch.int32)), "sharded ld_area values"
        assert got.n_pairs == want.n_pairs, "sharded ld_area pair count"
        own = ldist.ld_area_sharded(whole, pos, queries, flank, measure, thres, gather=False)
        n_own = torch.tensor([len(own)], dtype=torch.int64)
        dist.all_reduce(n_own)
        assert int(n_own) == len(want)
    # the WORK of the matrix-pipe band splits with the query ranges: a rank evaluates the passes its own queries can be
    # part of (its share of the band + a halo of one window), not the whole band
    n_snps, n_hap = 20000, 1008
    whole = PackedPanel.from_codes(synth.synth_codes_device(n_snps, n_hap, seed=9, device=dev))
    pos = synth.synth_positions(n_snps, step=500)
    want = ld_area(whole, pos, None, 250000, "r_square", 0.8)
    b, e = ldist.query_partition(pos, None, 250000, world)[rank]
    mine = ld_area(whole, pos, list(range(b, e)), 250000, "r_square", 0.8)
    assert mine.band_passes is not None and want.band_passes is not None
    assert mine.band_passes <= 0.56 * want.band_passes, (mine.band_passes, want.band_passes)
    tot = torch.tensor([mine.band_passes], dtype=torch.int64)
    dist.all_reduce(tot)
    assert want.band_passes <= int(tot) <= 1.12 * want.band_passes, (int(tot), want.band_passes)
    got = ldist.ld_area_sharded(whole, pos, None, 250000, "r_square", 0.8)
    assert torch.equal(got.query, want.query) and torch.equal(got.oppos, want.oppos)
    assert torch.equal(got.ld32.view(torch.int32), want.ld32.view(torch.int32))
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        print("GPU_DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
