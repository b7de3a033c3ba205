"""Multi-GPU sharding of the ld_triangle pair loop and the ld_area window loop: one process per GPU, torch.distributed.

The pair matrix has no cross-pair dependency (every cell needs rows i and j only), so the work
shards without a data-path reduction.  What IS exchanged is the packed panel: each rank packs
the SNP rows it ingested (rows [r*N/R, (r+1)*N/R) rounded to whole 128-row slabs), then one
all-gather of the slab images (8 MB per rank at 100k x 5008 over 8 GPUs) and of the per-SNP
count vectors gives every rank the full ALT plane.  Slab images are contiguous in the tiled
layout, so the gathered buffer IS the full plane -- no re-layout.  Each rank then computes a
contiguous, equal share of the triangle's work units (equal pair counts by construction; unit
u's results live at (u - unit_begin) * 1024 of the rank's own output).

ld_area shards by query: every rank scans a contiguous range of the query list against the full panel
(query_partition balances window populations) and the sparse hit lists are all-gathered afterwards (gather_hits).

Backend "nccl" is RCCL over xGMI on ROCm; "gloo" (CPU tensors) is used by the tests of the
partition logic.
"""
from __future__ import annotations

from typing import List, Tuple

SLAB = 128
GROUP = 8


def n_chunks(n_hap: int) -> int:
    """Same arithmetic as ldx_n_chunks (include/ldx.h): 128-haplotype chunks per row, allocated in pairs."""
    return (n_hap + 255) // 256 * 2


def slab_partition(n_snps: int, world: int) -> List[Tuple[int, int]]:
    """Rows [begin, end) packed by each rank: whole slabs, ceil(n_slabs / world) per rank until they run out
    (only the last ranks get fewer, possibly none), covering [0, n_snps).  With this shape the rank-major
    concatenation of equally padded shards IS the full tiled plane followed by padding, so the exchange needs no
    per-rank placement (fused_gather)."""
    n_slabs = (n_snps + SLAB - 1) // SLAB
    big = (n_slabs + world - 1) // world
    out = []
    for r in range(world):
        s0 = min(n_slabs, r * big)
        s1 = min(n_slabs, (r + 1) * big)
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


def pairs_in_units(n_snps: int, u0: int, u1: int) -> int:
    """Number of valid cells (row > col, row < n_snps) -- i.e. SNP pairs -- in units [u0, u1): a rank's share of the work
    (bench.py prints it per rank; equal UNIT counts are equal pair counts up to the diagonal and the padding)."""
    import numpy as np

    t_count = (n_snps + SLAB - 1) // SLAB
    G = t_count * (SLAB // GROUP)
    total = 0
    for t in range(t_count):
        base = t * G - 8 * t * (t - 1)
        nxt = (t + 1) * G - 8 * (t + 1) * t
        a, b = max(u0, base), min(u1, nxt)
        if a >= b:
            continue
        g = np.arange(a, b, dtype=np.int64) - base + 16 * t
        r = (g[:, None] * GROUP + np.arange(GROUP, dtype=np.int64)[None, :]).ravel()
        r = r[r < n_snps]
        total += int(np.clip(r - t * SLAB, 0, SLAB).sum())
    return total


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


def _gather_layout(slabs, slab_bytes: int, with_ref: bool):
    big = max(slabs)
    cnt_bytes = SLAB * 4
    per_slab = slab_bytes * (2 if with_ref else 1) + 2 * cnt_bytes
    o_acnt = big * slab_bytes
    o_rcnt = o_acnt + big * cnt_bytes
    o_ref = o_rcnt + big * cnt_bytes
    return big, cnt_bytes, per_slab, o_acnt, o_rcnt, o_ref


def fused_gather_start(local, slabs, slab_bytes: int, dev, group=None, stage=None, with_ref: bool = False,
                       async_op: bool = False):
    """First half of fused_gather: fill this rank's byte shard and issue the ONE all-gather.  Returns
    (stage, work): the (send, recv) staging buffers and, with ``async_op``, the collective's Work handle (else None:
    the call has already made the current stream wait for the collective)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    big, cnt_bytes, per_slab, o_acnt, o_rcnt, o_ref = _gather_layout(slabs, slab_bytes, with_ref)
    if stage is None or stage[0].numel() != big * per_slab or stage[1].numel() != world * big * per_slab:
        stage = (torch.zeros(big * per_slab, dtype=torch.uint8, device=dev),
                 torch.empty(world * big * per_slab, dtype=torch.uint8, device=dev))
    send, recv = stage
    fill_shard(send, local, slabs[rank], slabs, slab_bytes, with_ref)
    work = dist.all_gather_into_tensor(recv, send, group=group, async_op=async_op)
    return stage, (work if async_op else None)


def fill_shard(send, local, mine: int, slabs, slab_bytes: int, with_ref: bool) -> None:
    """One rank's byte shard of the fused exchange: [ALT: big slabs | acnt | rcnt (| REF: big slabs)] with this rank's
    ``mine`` slabs at the front of every piece (the rest of a piece is padding nobody reads)."""
    import torch

    _, cnt_bytes, _, o_acnt, o_rcnt, o_ref = _gather_layout(slabs, slab_bytes, with_ref)
    if local is not None and mine:
        send[: mine * slab_bytes] = local["alt"][: mine * slab_bytes]
        send[o_acnt: o_acnt + mine * cnt_bytes] = local["acnt"][: mine * SLAB].view(torch.uint8)
        send[o_rcnt: o_rcnt + mine * cnt_bytes] = local["rcnt"][: mine * SLAB].view(torch.uint8)
        if with_ref:
            send[o_ref: o_ref + mine * slab_bytes] = local["ref"][: mine * slab_bytes]


def fused_gather_finish(full, stage, slabs, slab_bytes: int, work=None):
    """Second half: wait for the collective (if it was issued asynchronously) and copy the pieces of every rank's
    shard to their places in ``full`` (strided copies when the shards are equal)."""
    import torch

    if work is not None:
        work.wait()                                # the current stream waits; the host does not
    with_ref = "ref" in full
    world = len(slabs)
    big, cnt_bytes, per_slab, o_acnt, o_rcnt, o_ref = _gather_layout(slabs, slab_bytes, with_ref)
    shard = stage[1].view(world, big * per_slab)
    acnt8, rcnt8 = full["acnt"].view(torch.uint8), full["rcnt"].view(torch.uint8)
    if len(set(slabs)) == 1:                       # equal shards: one strided copy per piece
        full["alt"].view(world, big * slab_bytes).copy_(shard[:, :o_acnt])
        acnt8.view(world, big * cnt_bytes).copy_(shard[:, o_acnt:o_rcnt])
        rcnt8.view(world, big * cnt_bytes).copy_(shard[:, o_rcnt:o_ref])
        if with_ref:
            full["ref"].view(world, big * slab_bytes).copy_(shard[:, o_ref:])
    elif all(n == big for n in slabs[: max(1, (sum(slabs) + big - 1) // big) - 1]):
        # slab_partition's shape (every rank but the last non-empty one holds `big` slabs): the rank-major
        # concatenation of a piece is the full piece followed by padding -- two copies per piece, no per-rank loop
        full["alt"].copy_(shard[:, :o_acnt].reshape(-1)[: full["alt"].numel()])
        acnt8.copy_(shard[:, o_acnt:o_rcnt].reshape(-1)[: acnt8.numel()])
        rcnt8.copy_(shard[:, o_rcnt:o_ref].reshape(-1)[: rcnt8.numel()])
        if with_ref:
            full["ref"].copy_(shard[:, o_ref:].reshape(-1)[: full["ref"].numel()])
    else:                                          # arbitrary shard sizes: place rank by rank
        off = 0
        for r, n in enumerate(slabs):
            full["alt"][off * slab_bytes: (off + n) * slab_bytes] = shard[r, : n * slab_bytes]
            acnt8[off * cnt_bytes: (off + n) * cnt_bytes] = shard[r, o_acnt: o_acnt + n * cnt_bytes]
            rcnt8[off * cnt_bytes: (off + n) * cnt_bytes] = shard[r, o_rcnt: o_rcnt + n * cnt_bytes]
            if with_ref:
                full["ref"][off * slab_bytes: (off + n) * slab_bytes] = shard[r, o_ref: o_ref + n * slab_bytes]
            off += n


def fused_gather(full, local, slabs, slab_bytes: int, group=None, stage=None):
    """ONE all-gather for a whole panel shard.  ``full`` / ``local`` are dicts of 1-D tensors (any device /
    backend): 'alt' (uint8, whole slab images), 'acnt', 'rcnt' (int32, 128 per slab) and optionally 'ref'; ``local``
    holds this rank's slabs (or is None), ``full`` receives all ranks' in rank order; ``slabs[r]`` = slabs of rank r.

    Every rank contributes a single byte shard [ALT: big slabs | acnt | rcnt (| REF: big slabs)], big = the largest
    rank's slab count; the pieces are then copied to their places (strided copies when the shards are equal).
    Returns the (send, recv) staging buffers for re-use.
    """
    stage, _ = fused_gather_start(local, slabs, slab_bytes, full["alt"].device, group, stage, "ref" in full)
    fused_gather_finish(full, stage, slabs, slab_bytes)
    return stage


def all_gather_panel(local, n_snps: int, n_hap: int, group=None, out=None, with_ref: bool = False):
    """All-gather per-rank slab shards into a full PackedPanel (device tensors, RCCL).

    ``local`` is the PackedPanel of this rank's rows (slab_partition(n_snps, world)[rank]), or None
    when the rank owns no rows; its planes are whole slab images, so concatenating the ranks' planes
    in rank order yields the tiled plane of the full panel.  One collective per call (fused_gather): a ring
    all-gather over xGMI is latency-bound at these sizes (8 MB per rank at 100k x 5008 over 8 GPUs), so one
    call instead of four is what matters.  The REF plane only feeds the per-SNP counts, which travel anyway,
    so it is gathered only on request.  ``out`` re-uses a full panel (and its staging buffers).
    """
    import torch.distributed as dist

    from .panel import PackedPanel, require_gpu

    world = dist.get_world_size(group)
    parts = slab_partition(n_snps, world)
    dev = local.device if local is not None else require_gpu()
    full = out if out is not None else PackedPanel.empty(n_snps, n_hap, dev)
    slab_bytes = n_chunks(n_hap) * SLAB * 16
    slabs = [(e - b + SLAB - 1) // SLAB for (b, e) in parts]
    fd = {"alt": full.alt, "acnt": full.acnt, "rcnt": full.rcnt}
    ld = None if local is None else {"alt": local.alt, "acnt": local.acnt, "rcnt": local.rcnt}
    if with_ref:
        fd["ref"] = full.ref
        if ld is not None:
            ld["ref"] = local.ref
    full._gather_stage = fused_gather(fd, ld, slabs, slab_bytes, group, getattr(full, "_gather_stage", None))
    full.refresh_stats()
    return full


class PanelPipeline:
    """Double-buffered exchange for a stream of batches: while the kernel of batch k runs, the all-gather of batch
    k + 1 is in flight (RCCL runs collectives on its own stream; only the Work's wait() ties it back to the compute
    stream).  Per batch:  ``start(local)`` fills the send shard and issues the all-gather asynchronously,
    ``finish()`` -- one batch later -- makes the compute stream wait for it, places the pieces in the next of two
    full panels and returns that panel.  At most one exchange is in flight; buffers alternate, so the panel a kernel
    is still reading is never the one being filled."""

    def __init__(self, n_snps: int, n_hap: int, device, group=None):
        import torch.distributed as dist

        from .panel import PackedPanel

        self.n_snps, self.n_hap, self.group = n_snps, n_hap, group
        self.world = dist.get_world_size(group)
        parts = slab_partition(n_snps, self.world)
        self.slabs = [(e - b + SLAB - 1) // SLAB for (b, e) in parts]
        self.slab_bytes = n_chunks(n_hap) * SLAB * 16
        self.panels = [PackedPanel.empty(n_snps, n_hap, device) for _ in range(2)]
        self.stages = [None, None]
        self.device = device
        self.issued = 0          # exchanges started
        self.done = 0            # exchanges finished
        self.work = None

    def start(self, local) -> None:
        if self.issued != self.done:
            raise RuntimeError("PanelPipeline.start: the previous exchange has not been finished")
        k = self.issued % 2
        ld = None if local is None else {"alt": local.alt, "acnt": local.acnt, "rcnt": local.rcnt}
        self.stages[k], self.work = fused_gather_start(ld, self.slabs, self.slab_bytes, self.device, self.group,
                                                       self.stages[k], False, async_op=True)
        self.issued += 1

    def finish(self):
        if self.issued != self.done + 1:
            raise RuntimeError("PanelPipeline.finish: no exchange in flight")
        k = self.done % 2
        full = self.panels[k]
        fused_gather_finish({"alt": full.alt, "acnt": full.acnt, "rcnt": full.rcnt}, self.stages[k], self.slabs,
                            self.slab_bytes, self.work)
        self.work = None
        self.done += 1
        full.refresh_stats()
        return full


# --------------------------------------------------------------------------- ld_area
def query_partition(positions, queries, flank: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous ranges [begin, end) of the ASCENDING query list per rank, cut so that every rank gets a near-equal
    number of (query, opposing) pairs: the cost of a query is its window population (ld_area.py:174-177,215-217),
    which varies with the local SNP density, so equal query counts would not balance.  Ranges tile [0, n_query) in
    rank order, hence the concatenation of the ranks' hit lists in rank order is the single-process hit list."""
    import numpy as np

    pos = np.asarray(positions, dtype=np.int64)
    q = np.arange(pos.size, dtype=np.int64) if queries is None else np.unique(np.asarray(queries, dtype=np.int64))
    if q.size == 0:
        return [(0, 0)] * world
    qpos = pos[q]
    lo = np.searchsorted(pos, np.maximum(qpos - flank, 0), side="right")
    hi = np.searchsorted(pos, qpos + flank, side="right")
    cost = np.cumsum((hi - lo).astype(np.int64) + 1)                 # +1: a query with an empty window still costs a visit
    cuts = [0]
    for r in range(1, world):
        cuts.append(max(cuts[-1], int(np.searchsorted(cost, cost[-1] * r // world, side="left"))))
    cuts.append(int(q.size))
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def gather_hits(query, oppos, ld32, group=None):
    """All-gather the ranks' hit lists (variable length) in rank order: counts first, then ONE padded all-gather of
    16-byte records {query row, opposing row, r_square bits, d_prime bits}.  Works with device tensors (RCCL) and CPU
    tensors (gloo).  Returns (query int64 [n], oppos int64 [n], ld32 float32 [n, 2]) over all ranks."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = query.device
    n_mine = int(query.numel())
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, torch.tensor([n_mine], dtype=torch.int64, device=dev), group=group)
    sizes = [int(c) for c in counts.tolist()]
    rec = torch.empty((n_mine, 4), dtype=torch.int32, device=dev)
    rec[:, 0] = query.to(torch.int32)
    rec[:, 1] = oppos.to(torch.int32)
    rec[:, 2:4] = ld32.reshape(n_mine, 2).contiguous().view(torch.int32)
    total = sum(sizes)
    out = torch.empty(total * 4, dtype=torch.int32, device=dev)
    if total:
        gather_shards(out, rec.reshape(-1), [4 * s for s in sizes], group)
    out = out.view(total, 4)
    return out[:, 0].to(torch.int64), out[:, 1].to(torch.int64), out[:, 2:4].contiguous().view(torch.float32)


def ld_area_sharded(panel, positions, queries=None, flank: int = 100000, measure: str = "r_square", thres: float = 0.8,
                    group=None, gather: bool = True):
    """ld_area over all ranks: every rank holds the full panel (all_gather_panel) and scans ITS range of the query list
    (query_partition); no data-path exchange during the scan.  With ``gather`` the hit lists are then all-gathered
    (gather_hits) and every rank returns the complete AreaHits, identical to the single-GPU ld_area; without it each
    rank keeps its own hits (the writers of ld_area.py:261-292 are per query, so a rank can write its own files)."""
    import numpy as np
    import torch.distributed as dist

    from .ops import AreaHits, ld_area

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    pos = np.asarray(positions.cpu() if hasattr(positions, "cpu") else positions, dtype=np.int64)
    q = np.arange(pos.size, dtype=np.int64) if queries is None else np.unique(np.asarray(queries, dtype=np.int64))   # strictly ascending, like ops.ld_area
    b, e = query_partition(pos, q, flank, world)[rank]
    mine = ld_area(panel, positions, q[b:e].tolist(), flank, measure, thres)
    if not gather:
        return mine
    qa, oa, la = gather_hits(mine.query, mine.oppos, mine.ld32, group)
    lo = np.searchsorted(pos, np.maximum(pos[q] - flank, 0), side="right") if q.size else np.zeros(0, np.int64)
    hi = np.searchsorted(pos, pos[q] + flank, side="right") if q.size else np.zeros(0, np.int64)
    n_pairs = int((hi - lo).sum()) - (int((np.maximum(pos[q] - flank, 0) < pos[q]).sum()) if q.size else 0)
    return AreaHits(qa, oa, la, n_pairs)
