"""Multi-GPU sharding of the ld_triangle pair loop: one process per GPU, torch.distributed.

The pair matrix has no cross-pair dependency (every cell needs rows i and j only), so the work
shards without a data-path reduction.  What IS exchanged is the packed panel: each rank packs
the SNP rows it ingested (rows [r*N/R, (r+1)*N/R) rounded to whole 128-row slabs), then one
all-gather of the slab images (8 MB per rank at 100k x 5008 over 8 GPUs) and of the per-SNP
count vectors gives every rank the full ALT plane.  Slab images are contiguous in the tiled
layout, so the gathered buffer IS the full plane -- no re-layout.  Each rank then computes a
contiguous, equal share of the triangle's work units (equal pair counts by construction; unit
u's results live at (u - unit_begin) * 1024 of the rank's own output).

Backend "nccl" is RCCL over xGMI on ROCm; "gloo" (CPU tensors) is used by the tests of the
partition logic.
"""
from __future__ import annotations

from typing import List, Tuple

SLAB = 128
GROUP = 8


def slab_partition(n_snps: int, world: int) -> List[Tuple[int, int]]:
    """Rows [begin, end) packed by each rank: whole slabs, as even as possible, covering [0, n_snps)."""
    n_slabs = (n_snps + SLAB - 1) // SLAB
    out = []
    for r in range(world):
        s0 = n_slabs * r // world
        s1 = n_slabs * (r + 1) // world
        out.append((min(n_snps, s0 * SLAB), min(n_snps, s1 * SLAB)))
    return out


def triangle_units(n_snps: int) -> int:
    """Same arithmetic as ldx_triangle_units (include/ldx.h)."""
    t = (n_snps + SLAB - 1) // SLAB
    g = t * (SLAB // GROUP)
    return t * g - 8 * t * (t - 1)


def unit_partition(n_snps: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous unit ranges [begin, end) per rank: equal work within one unit."""
    total = triangle_units(n_snps)
    return [(total * r // world, total * (r + 1) // world) for r in range(world)]


def unit_cells(n_snps: int, u0: int, u1: int):
    """(rows, cols) int64 arrays of the valid cells (row > col, row < n_snps) of units [u0, u1)."""
    import numpy as np

    t_count = (n_snps + SLAB - 1) // SLAB
    G = t_count * (SLAB // GROUP)
    rows, cols = [], []
    for t in range(t_count):
        base = t * G - 8 * t * (t - 1)
        nxt = (t + 1) * G - 8 * (t + 1) * t
        a, b = max(u0, base), min(u1, nxt)
        if a >= b:
            continue
        g = np.arange(a, b, dtype=np.int64) - base + 16 * t
        r = (g[:, None] * GROUP + np.arange(GROUP, dtype=np.int64)[None, :]).ravel()
        c = t * SLAB + np.arange(SLAB, dtype=np.int64)
        rr, cc = np.meshgrid(r, c, indexing="ij")
        m = (rr > cc) & (rr < n_snps)
        rows.append(rr[m])
        cols.append(cc[m])
    if not rows:
        z = np.zeros(0, dtype=np.int64)
        return z, z
    return np.concatenate(rows), np.concatenate(cols)


def gather_shards(dst, src, sizes, group=None):
    """All-gather variable-size 1-D shards (rank r contributes src[:sizes[r]]) into dst, in rank order.

    Works on any backend (device tensors with nccl/RCCL, CPU tensors with gloo).  Equal shards are
    gathered straight into ``dst``; uneven ones are padded to the largest (collectives need equal
    shapes) and copied out.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(sizes) == world
    mine = src[: sizes[rank]]
    total = sum(sizes)
    if len(set(sizes)) == 1:
        dist.all_gather_into_tensor(dst[:total], mine.contiguous(), group=group)
        return
    big = max(sizes)
    send = torch.zeros(big, dtype=dst.dtype, device=dst.device)
    send[: mine.numel()] = mine
    recv = torch.empty(world * big, dtype=dst.dtype, device=dst.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    off = 0
    for r, n in enumerate(sizes):
        dst[off:off + n] = recv[r * big: r * big + n]
        off += n


def all_gather_panel(local, n_snps: int, n_hap: int, group=None):
    """All-gather per-rank slab shards into a full PackedPanel (device tensors, RCCL).

    ``local`` is the PackedPanel of this rank's rows (slab_partition(n_snps, world)[rank]), or None
    when the rank owns no rows; its planes are whole slab images, so concatenating the ranks' planes
    in rank order yields the tiled plane of the full panel.
    """
    import torch
    import torch.distributed as dist

    from .panel import PackedPanel, require_gpu

    world = dist.get_world_size(group)
    parts = slab_partition(n_snps, world)
    dev = local.device if local is not None else require_gpu()
    full = PackedPanel.empty(n_snps, n_hap, dev)
    slab_bytes = ((n_hap + 127) // 128) * SLAB * 16
    slabs = [(e - b + SLAB - 1) // SLAB for (b, e) in parts]

    def gather(dst, src, per_slab):
        if src is None:
            src = torch.empty(0, dtype=dst.dtype, device=dev)
        gather_shards(dst, src, [n * per_slab for n in slabs], group)

    gather(full.alt, None if local is None else local.alt, slab_bytes)
    gather(full.ref, None if local is None else local.ref, slab_bytes)
    gather(full.acnt, None if local is None else local.acnt, SLAB)
    gather(full.rcnt, None if local is None else local.rcnt, SLAB)
    full.refresh_stats()
    return full
