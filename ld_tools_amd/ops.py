"""Batched LD operators over a PackedPanel -- the host-side mirror of the reference's pair loops.

    ld_triangle(panel)                      <- ld_triangle.py:133-230  (all row > col pairs)
    ld_area(panel, positions, queries, ...) <- ld_area.py:152-276      (windowed scan, thresholded hits)
    pair_counts(panel_i, panel_j)           <- calc_ld.py:32           (bit-exact n11 block)
    ld_from_counts(n, n11, a1, r1, a2, r2)  <- calc_ld.py:33-97        (the epilogue alone)

Every function enqueues HIP kernels of libldx.so on torch's current stream and returns device
tensors; nothing here computes LD on the host.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import MEASURES, UNIT_PAIRS, check, lib
from .panel import PackedPanel, _ptr, _stream_ptr


PATHS = {"auto": 0, "popcount": 1, "mfma": 2, "fp4": 3}


def set_triangle_path(name: str) -> None:
    """Choose the kernel behind ld_triangle: 'popcount', 'mfma' or 'auto' (results are identical)."""
    check(lib.ldx_set_triangle_path(PATHS[name]), "ldx_set_triangle_path")


def get_triangle_path() -> str:
    code = lib.ldx_get_triangle_path()
    return next(k for k, v in PATHS.items() if v == code)


def set_area_path(name: str) -> None:
    """Choose the kernel behind ld_area: 'popcount' (scan of the query rows), 'mfma' (the whole +-flank band on the
    matrix pipe) or 'auto' (mfma when at least 1/16 of the SNPs are queries).  The hit sets are identical."""
    check(lib.ldx_set_area_path(PATHS[name]), "ldx_set_area_path")


def get_area_path() -> str:
    code = lib.ldx_get_area_path()
    return next(k for k, v in PATHS.items() if v == code)


# --------------------------------------------------------------------------- triangle
@dataclass
class TriangleResult:
    """Strip-packed lower triangle of one panel (layout: include/ldx.h, "Triangle work units").

    The cells are in ONE of the formats of include/ldx.h: ``ld32`` (float32 nearest to k / 10^4, -0.0 = the
    reference's int 0, 8 bytes per pair), ``k16`` (k itself in 15 bits + an int-0 bit, 4 bytes per pair) or -- one measure
    only, 2 bytes per pair, what a table writer needs (ld_triangle.py:223-230,344-360 print ONE measure) -- ``k16r`` /
    ``k16d`` (the r_square / d_prime half of k16).  Values a format cannot hold (>= 1024 / >= 3.2767: only with missing
    codes) are escape cells; ``dense_values()`` resolves them through ldx_ld_pairs_dev, so nothing is ever returned inexact.
    """

    n_snps: int
    unit_begin: int
    unit_end: int
    ld32: Optional[torch.Tensor] = None      # float32 [(units)*1024, 2]  (r_square, d_prime) rounded to 4 dp
    raw: Optional[torch.Tensor] = None       # float64 [(units)*1024, 2]  unrounded
    n11: Optional[torch.Tensor] = None       # int32   [(units)*1024]     alt/alt haplotype counts
    k16: Optional[torch.Tensor] = None       # int16   [(units)*1024, 2]  bit patterns of the uint16 cells
    panel: Optional[PackedPanel] = None      # the panel the result came from (resolves escape cells)
    k16one: Optional[torch.Tensor] = None    # int16   [(units)*1024]     one measure's uint16 cells (fmt k16r / k16d)
    one_fmt: Optional[str] = None            # "k16r" / "k16d" when k16one is set
    ws: Optional[torch.Tensor] = None        # the matrix kernel's pass-scheduler workspace (include/ldx.h, ldx_triangle_ex_dev):
                                             # zeroed once here, re-armed by every launch; one per result buffer, because two
                                             # launches that may overlap write different buffers

    @property
    def fmt(self) -> str:
        if self.k16one is not None:
            return self.one_fmt
        return "ld32" if self.ld32 is not None else "k16"

    @property
    def cells(self) -> torch.Tensor:
        if self.k16one is not None:
            return self.k16one
        return self.ld32 if self.ld32 is not None else self.k16

    def cell_index(self, rows, cols) -> np.ndarray:
        """Flat element index (relative to this shard) of cells (row > col)."""
        rows = np.asarray(rows, dtype=np.int64)
        cols = np.asarray(cols, dtype=np.int64)
        if np.any(rows <= cols):
            raise ValueError("cell_index needs row > col")
        npad = lib.ldx_padded_snps(self.n_snps)
        G = npad // 8
        t = cols // 128
        g = rows // 8
        u = t * G - 8 * t * (t - 1) + (g - 16 * t)
        return (u - self.unit_begin) * UNIT_PAIRS + _lib.cell_offset(rows % 8, cols % 128, self.fmt)

    def k_and_int0(self, idx) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(k int64 [m, 2], int0 bool [m, 2], escape bool [m, 2]) of the cells at flat indices ``idx`` (host arrays);
        k of an escape cell is -1."""
        ix = torch.as_tensor(np.asarray(idx, dtype=np.int64), device=self.cells.device)
        if self.k16one is not None:          # one measure: arrays of shape [m, 1]
            u = self.k16one[ix].cpu().numpy().view(np.uint16).astype(np.int64).reshape(-1, 1)
            int0 = (u & _lib.K16_INT0) != 0
            esc = u == _lib.K16_BIG
            return np.where(esc, -1, np.where(int0, 0, u)), int0, esc
        if self.ld32 is not None:
            v = self.ld32[ix].cpu().numpy()
            esc = np.isnan(v)
            k = np.where(esc, -1, np.rint(np.nan_to_num(v).astype(np.float64) * 1e4)).astype(np.int64)
            return k, np.signbit(v) & (v == 0), esc
        u = self.k16[ix].cpu().numpy().view(np.uint16).astype(np.int64)
        int0 = (u & _lib.K16_INT0) != 0
        esc = u == _lib.K16_BIG
        return np.where(esc, -1, np.where(int0, 0, u)), int0, esc

    def dense(self, measure: str = "r_square", thres: Optional[float] = None,
              rows: Optional[Tuple[int, int]] = None) -> torch.Tensor:
        """ld_two_dim of ld_triangle.py:114,223-230 as float32 [rows][n_snps] on the device.

        Cells the reference leaves at the template's int 0 (row <= col, or rounded measure below
        ``thres``) hold -0.0; a computed int 0 (monomorphic variant) is -0.0 too, a float 0.0 is +0.0.
        Escape cells come out as NaN (``dense_values`` resolves them).  Needs the full triangle.
        """
        if self.unit_begin != 0 or self.unit_end != lib.ldx_triangle_units(self.n_snps):
            raise _lib.LdxError("dense() needs an unsharded TriangleResult")
        if self.k16one is not None and _lib.ONE_MEASURE[self.one_fmt] != measure:
            raise _lib.LdxError(f"this result holds {_lib.ONE_MEASURE[self.one_fmt]} only (format {self.one_fmt})")
        r0, r1 = rows if rows is not None else (0, self.n_snps)
        out = torch.empty((r1 - r0, self.n_snps), dtype=torch.float32, device=self.cells.device)
        check(lib.ldx_triangle_dense_ex_dev(self.cells.data_ptr(), _lib.FORMATS[self.fmt], self.n_snps,
                                            MEASURES[measure], 0 if thres is None else 1,
                                            0.0 if thres is None else float(thres), r0, r1, out.data_ptr(),
                                            self.n_snps, _stream_ptr()), "ldx_triangle_dense_ex_dev")
        return out

    def dense_values(self, measure: str = "r_square", thres: Optional[float] = None,
                     rows: Optional[Tuple[int, int]] = None):
        """``dense`` on the host plus the exact values of its escape cells: (float32 array [rows][n_snps],
        {(row, col): Python value}).  The dict holds, for every NaN cell of the array, what the reference's
        ld_two_dim holds there: round(x, 4) as a float, or the int 0 when it lies below ``thres``
        (ld_triangle.py:223-230)."""
        r0, _ = rows if rows is not None else (0, self.n_snps)
        d = self.dense(measure, thres, rows).cpu().numpy()
        rr, cc = np.nonzero(np.isnan(d))
        fixes = {}
        if rr.size:
            if self.panel is None:
                raise _lib.LdxError("escape cells need the panel (TriangleResult.panel) to be resolved")
            ex = ld_pairs(self.panel, rr + r0, cc)
            col = 0 if measure == "r_square" else 1
            for a, b, k in zip((rr + r0).tolist(), cc.tolist(), ex["k"][:, col].tolist()):
                v = k / 10000.0
                fixes[(a, b)] = 0 if (thres is not None and v < thres) else v
        return d, fixes


def ld_triangle(panel: PackedPanel, unit_range: Optional[Tuple[int, int]] = None, want_raw: bool = False,
                want_n11: bool = False, out: Optional[TriangleResult] = None, fmt: str = "ld32",
                path: Optional[str] = None) -> TriangleResult:
    """All row > col pairs of the panel: var_1 = row, var_2 = col (ld_triangle.py:193-194).

    ``unit_range`` restricts the work to a contiguous slice of the unit list (multi-GPU sharding);
    ``out`` re-uses the buffers of a previous result of the same shape (benchmark loops); ``fmt`` picks the cell
    format ('ld32': 8 bytes per pair, 'k16': 4 bytes per pair, 'k16r' / 'k16d': ONE measure, 2 bytes per pair -- the kernel
    then skips the other value's arithmetic); ``path`` overrides the process-wide kernel choice ('fp4', 'mfma', 'popcount')
    for this call.
    """
    total = panel.n_units
    u0, u1 = (0, total) if unit_range is None else unit_range
    u0, u1 = max(0, u0), min(total, u1)
    cells = max(0, u1 - u0) * UNIT_PAIRS
    dev = panel.device
    if fmt not in _lib.FORMATS:
        raise _lib.LdxError(f"unknown cell format {fmt!r}")
    if fmt != "ld32" and want_raw:
        raise _lib.LdxError("the unrounded output travels with the ld32 format only")
    if fmt in _lib.ONE_MEASURE and want_n11:
        raise _lib.LdxError("the one-measure formats take no side output")
    if out is None:
        out = TriangleResult(panel.n_snps, u0, u1,
                             torch.empty((cells, 2), dtype=torch.float32, device=dev) if fmt == "ld32" else None,
                             torch.empty((cells, 2), dtype=torch.float64, device=dev) if want_raw else None,
                             torch.empty(cells, dtype=torch.int32, device=dev) if want_n11 else None,
                             torch.empty((cells, 2), dtype=torch.int16, device=dev) if fmt == "k16" else None,
                             k16one=torch.empty(cells, dtype=torch.int16, device=dev) if fmt in _lib.ONE_MEASURE else None,
                             one_fmt=fmt if fmt in _lib.ONE_MEASURE else None)
    elif (out.n_snps, out.unit_begin, out.unit_end, out.fmt) != (panel.n_snps, u0, u1, fmt):
        raise _lib.LdxError("ld_triangle: `out` has a different shape or format")
    out.panel = panel
    if cells:
        pcode = lib.ldx_get_triangle_path() if path is None else PATHS[path]
        if out.ws is None and pcode != PATHS["popcount"]:
            out.ws = torch.zeros(lib.ldx_triangle_workspace_bytes(), dtype=torch.uint8, device=dev)
        check(lib.ldx_triangle_ex_dev(panel.alt.data_ptr(), panel.fa.data_ptr(), panel.fr.data_ptr(),
                                      panel.q.data_ptr(), panel.n_snps, panel.n_hap, u0, u1, pcode,
                                      _lib.FORMATS[fmt], out.cells.data_ptr(), _ptr(out.raw), _ptr(out.n11),
                                      _ptr(out.ws), 0 if out.ws is None else out.ws.numel(), _stream_ptr()),
              "ldx_triangle_ex_dev")
    return out


# --------------------------------------------------------------------------- n11 block / epilogue / explicit pairs
def pair_counts(panel_i: PackedPanel, panel_j: Optional[PackedPanel] = None) -> torch.Tensor:
    """n11[i][j] = #haplotypes with code 1 at SNP i of panel_i and SNP j of panel_j (calc_ld.py:32)."""
    pj = panel_j or panel_i
    if pj.n_hap != panel_i.n_hap:
        raise _lib.LdxError("pair_counts: panels differ in haplotype count")
    out = torch.empty((panel_i.n_snps, pj.n_snps), dtype=torch.int32, device=panel_i.device)
    check(lib.ldx_pair_counts_dev(panel_i.alt.data_ptr(), panel_i.n_snps, pj.alt.data_ptr(), pj.n_snps,
                                  panel_i.n_hap, out.data_ptr(), pj.n_snps, _stream_ptr()),
          "ldx_pair_counts_dev")
    return out


def ld_pairs(panel: PackedPanel, rows, cols) -> dict:
    """calc_ld for an explicit list of pairs (var_1 = rows[p], var_2 = cols[p]) of one panel, exact for any magnitude:
    host arrays ``k`` (float64 [m, 2]: round(x, 4) * 10^4 of r_square, d_prime), ``raw`` (float64 [m, 2]), ``flags``
    (uint8, LDX_FLAG_*: which value is the reference's int 0) and ``n11`` (uint32)."""
    dev = panel.device
    r = torch.as_tensor(np.ascontiguousarray(np.asarray(rows, dtype=np.int64).astype(np.int32))).to(dev)
    c = torch.as_tensor(np.ascontiguousarray(np.asarray(cols, dtype=np.int64).astype(np.int32))).to(dev)
    m = int(r.numel())
    if int(c.numel()) != m:
        raise _lib.LdxError("ld_pairs: rows and cols differ in length")
    k = torch.empty((m, 2), dtype=torch.float64, device=dev)
    raw = torch.empty((m, 2), dtype=torch.float64, device=dev)
    flags = torch.empty(m, dtype=torch.uint8, device=dev)
    n11 = torch.empty(m, dtype=torch.int32, device=dev)
    if m:
        check(lib.ldx_ld_pairs_dev(panel.alt.data_ptr(), panel.acnt.data_ptr(), panel.rcnt.data_ptr(), panel.n_snps,
                                   panel.n_hap, r.data_ptr(), c.data_ptr(), m, k.data_ptr(), raw.data_ptr(),
                                   flags.data_ptr(), n11.data_ptr(), _stream_ptr()), "ldx_ld_pairs_dev")
    return {"k": k.cpu().numpy(), "raw": raw.cpu().numpy(), "flags": flags.cpu().numpy(),
            "n11": n11.cpu().numpy().view(np.uint32)}


def ld_from_counts(n: int, n11, a1, r1, a2, r2, device: Optional[torch.device] = None, full: bool = False):
    """The epilogue alone (calc_ld.py:33-97) on arrays of counts.

    Returns (raw float64 [m,2], rounded float32 [m,2], flags uint8 [m]) as device tensors; with ``full`` also
    k (float64 [m,2]: round(x, 4) * 10^4, exact for any magnitude), the 4-byte cells (int16 [m,2], the bit patterns
    of ldx_k16) and a bool [m] telling which pairs the fp32 epilogue tier would have kept (the others go to fp64).
    """
    dev = device or torch.device("cuda", torch.cuda.current_device())
    def up(x):
        t = torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.uint32)).view(np.int32))
        return t.to(dev)
    t11, ta1, tr1, ta2, tr2 = (up(x) for x in (n11, a1, r1, a2, r2))
    m = t11.numel()
    raw = torch.empty((m, 2), dtype=torch.float64, device=dev)
    rnd = torch.empty((m, 2), dtype=torch.float32, device=dev)
    flags = torch.empty(m, dtype=torch.uint8, device=dev)
    k = torch.empty((m, 2), dtype=torch.float64, device=dev) if full else None
    k16 = torch.empty((m, 2), dtype=torch.int16, device=dev) if full else None
    check(lib.ldx_ld_from_counts_ex_dev(int(n), m, t11.data_ptr(), ta1.data_ptr(), tr1.data_ptr(), ta2.data_ptr(),
                                        tr2.data_ptr(), raw.data_ptr(), _ptr(k), rnd.data_ptr(), _ptr(k16),
                                        flags.data_ptr(), _stream_ptr()), "ldx_ld_from_counts_ex_dev")
    sure32 = (flags & 0x80) != 0       # LDX_FLAG_F32_SURE: the fp32 epilogue tier would have kept the pair
    flags = flags & 3
    if full:
        return raw, rnd, flags, k, k16, sure32
    return raw, rnd, flags


# --------------------------------------------------------------------------- area
class AreaHits:
    """Thresholded hits of a windowed scan, sorted by (query row, opposing row) = VCF order."""

    def __init__(self, query: torch.Tensor, oppos: torch.Tensor, ld32: torch.Tensor, n_pairs, offsets=None,
                 band_passes=None):
        self.query = query        # int64 [n]  panel row of var_1 (the query)
        self.oppos = oppos        # int64 [n]  panel row of var_2 (the opposing variant)
        self.ld32 = ld32          # float32 [n, 2]  rounded (r_square, d_prime), -0.0 = int 0
        self._n_pairs = n_pairs   # int, or a callable that counts on first use (bookkeeping only)
        self.offsets = offsets    # int32 [n_snps + 1] (device): the hits of query row q are [offsets[q], offsets[q + 1])
        self._band_passes = band_passes

    @property
    def n_pairs(self) -> int:
        """(query, opposing) pairs inside the windows that were evaluated."""
        if callable(self._n_pairs):
            self._n_pairs = self._n_pairs()
        return self._n_pairs

    @property
    def band_passes(self) -> Optional[int]:
        """Passes (4 units of 64 rows x 128 columns) the matrix-pipe band evaluated for this call; None for the popcount
        scan.  Instrumentation: shows how the work of a sharded scan splits."""
        if callable(self._band_passes):
            self._band_passes = self._band_passes()
        return self._band_passes

    def __len__(self) -> int:
        return int(self.query.numel())

    def python_values(self, panel: PackedPanel):
        """[(r_square, d_prime)] per hit as the reference's Python values (float round(x, 4), or the int 0).  Hits whose
        float32 is the escape NaN (a value >= 1024: only with missing codes) are resolved exactly through ld_pairs."""
        v = self.ld32.cpu().numpy().reshape(-1, 2)
        esc = np.isnan(v)
        k = np.rint(np.where(esc, 0, v).astype(np.float64) * 1e4)
        int0 = np.signbit(v) & (v == 0)
        out = [[0 if z else kk / 10000.0 for kk, z in zip(kr.tolist(), zr.tolist())] for kr, zr in zip(k, int0)]
        rows = np.flatnonzero(esc.any(axis=1))
        if rows.size:
            ex = ld_pairs(panel, self.query.cpu().numpy()[rows], self.oppos.cpu().numpy()[rows])
            for r, kk in zip(rows.tolist(), ex["k"]):
                for c in (0, 1):
                    if esc[r, c]:
                        out[r][c] = float(kk[c]) / 10000.0
        return [tuple(x) for x in out]


class _AreaPlan:
    """Buffers of one ld_area call shape (panel, positions tensor, queries, flank, measure, threshold) and -- once the
    shape has been seen twice -- the HIP graph of its launches (query mask, band plan, scan, offsets, scatter, ordering:
    ten small launches around one 0.36 ms kernel, launch-bound when issued one by one)."""

    def __init__(self, panel, nq, cap):
        dev = panel.device
        self.cap = cap
        self.ws_bytes = lib.ldx_area_workspace_bytes(panel.n_snps, panel.n_hap, nq)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.fin_bytes = lib.ldx_area_finish_workspace_bytes(panel.n_snps)
        self.fin = torch.empty(self.fin_bytes, dtype=torch.uint8, device=dev)
        self.n_hits = torch.zeros(1, dtype=torch.int64, device=dev)
        self.summary = torch.zeros(2, dtype=torch.int64, device=dev)
        self.offsets = torch.empty(panel.n_snps + 1, dtype=torch.int32, device=dev)
        self.raw = torch.empty((cap, 4), dtype=torch.int32, device=dev)      # ldx_hit = {u32, u32, f32, f32}
        self.hits = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        self.graph = None
        self.uses = 0
        self.nbytes = sum(int(t.numel()) * t.element_size() for t in (self.ws, self.fin, self.raw, self.hits, self.offsets))


_PLAN_BYTES_MAX = 1 << 30      # kept ld_area plans per panel (buffers a plan pins while it is kept)


def _area_launch(panel, pos, q, nq, flank, measure, thres, plan, events=None):
    """The launches of one scan + finish on torch's current stream (no host synchronisation)."""
    if events is not None:
        del events[:]
        events.extend(torch.cuda.Event(enable_timing=True) for _ in range(3))
        events[0].record()
    counts = lib.ldx_area_finish_counts(plan.fin.data_ptr())     # the scan counts per query row as it stores the hits
    check(lib.ldx_area_scan_dev(panel.alt.data_ptr(), panel.fa.data_ptr(), panel.fr.data_ptr(), panel.q.data_ptr(),
                                panel.n_snps, panel.n_hap, pos.data_ptr(), q.data_ptr(), nq, int(flank),
                                MEASURES[measure], float(thres), plan.raw.data_ptr(), plan.cap, plan.n_hits.data_ptr(),
                                counts, plan.ws.data_ptr(), plan.ws_bytes, _stream_ptr()), "ldx_area_scan_dev")
    if events is not None:
        events[1].record()
    check(lib.ldx_area_finish_ex_dev(plan.raw.data_ptr(), plan.n_hits.data_ptr(), plan.cap, panel.n_snps,
                                     plan.hits.data_ptr(), plan.offsets.data_ptr(), plan.summary.data_ptr(),
                                     plan.fin.data_ptr(), plan.fin_bytes, 1, _stream_ptr()), "ldx_area_finish_ex_dev")
    if events is not None:
        events[2].record()


def ld_area(panel: PackedPanel, positions, queries: Optional[Sequence[int]] = None, flank: int = 100000,
            measure: str = "r_square", thres: float = 0.8, hit_capacity: Optional[int] = None,
            check_positions: bool = True, events: Optional[list] = None, use_graph: Optional[bool] = None) -> AreaHits:
    """Windowed scan of ld_area.py:152-276 over the panel.

    positions: ascending 1-based coordinates of the panel's SNPs (VCF order).  queries: panel
    row indices of the query variants (default: every SNP).  For each query q the opposing
    variants are o != q with max(0, pos_q - flank) < pos_o <= pos_q + flank (pysam's fetch,
    ld_area.py:174-177,215-217); var_1 = query, var_2 = opposing; kept when the rounded
    ``measure`` >= thres (ld_area.py:248).

    Everything up to the ordered hit list runs on the device (scan, which counts per query as it stores -> exclusive scan
    -> scatter -> per-query order: ldx_area_scan_dev + ldx_area_finish_ex_dev); the host reads two integers at the end
    to size the result.  When the same scan shape comes again -- the same panel, the same DEVICE tensor of positions,
    the same queries / flank / measure / threshold: a driver walking tables of one chromosome -- its launches are
    replayed as ONE HIP graph from the second repetition on (``use_graph``: None = that rule, False = never, True = from the
    first call); the plan (buffers + graph) lives on the panel and keeps its buffers resident until it is evicted (eight
    shapes / 1 GiB per panel) or ``panel.clear_area_plans()`` is called.  A plan's graph carries its own ticket counters
    (in the plan's workspace), so plans of different panels or shapes may be replayed on different streams at once; one
    plan is one set of buffers -- the SAME shape on the same panel from two threads at once is the caller's to serialise.
    ``events`` (instrumentation, bench.py): a list that receives three torch events of the current stream -- before the
    scan, between the scan and the finishing kernels, after them; forces eager launches.
    """
    dev = panel.device
    pos_key = None
    if isinstance(positions, torch.Tensor):
        pos = positions.to(dev, dtype=torch.int64).contiguous()
        if pos.data_ptr() == positions.data_ptr():
            pos_key = (pos.data_ptr(), int(pos.numel()))
        # a device tensor is checked on the device (one more host round trip); callers that scan one chromosome many
        # times pass check_positions=False after the first call
        if check_positions and pos.numel() > 1 and bool((pos[1:] < pos[:-1]).any().item()):
            raise _lib.LdxError("positions must ascend (VCF order): the window search is a binary search")
    else:
        pos_h = np.ascontiguousarray(np.asarray(positions, dtype=np.int64))
        if pos_h.size > 1 and bool((pos_h[1:] < pos_h[:-1]).any()):
            raise _lib.LdxError("positions must ascend (VCF order): the window search is a binary search")
        pos = torch.as_tensor(pos_h).to(dev)
    if pos.numel() != panel.n_snps:
        raise _lib.LdxError("positions must have one entry per SNP")
    plans = panel.__dict__.setdefault("_area_plans", {})
    q_key = None
    if queries is None:
        q = plans.get("all_rows")
        if q is None:
            q = plans["all_rows"] = torch.arange(panel.n_snps, dtype=torch.int32, device=dev)
        q_key = "all"
    else:
        # the kernels want STRICTLY ascending rows (include/ldx.h): a repeated query is one query -- the reference would write
        # the same result file twice (ld_area.py:152-292) -- and "as many queries as SNPs" must mean every SNP once
        qn = np.unique(np.asarray(queries, dtype=np.int64))
        if qn.size == 0:
            e = torch.empty(0, dtype=torch.int64, device=dev)
            return AreaHits(e, e, torch.empty((0, 2), dtype=torch.float32, device=dev), 0,
                            torch.zeros(panel.n_snps + 1, dtype=torch.int32, device=dev))
        if qn[0] < 0 or qn[-1] >= panel.n_snps:
            raise _lib.LdxError("query row out of range")
        q = torch.as_tensor(qn.astype(np.int32)).to(dev)
    nq = int(q.numel())
    cap = int(hit_capacity) if hit_capacity is not None else max(1 << 20, 16 * nq)   # slots (16 B each); an overflow re-runs with the exact count
    # the plan of this call shape (only shapes that can come again unchanged have one that is kept)
    key = None
    if pos_key is not None and q_key is not None and events is None and use_graph is not False:
        key = (pos_key, q_key, int(flank), measure, float(thres), get_area_path(), panel.alt.data_ptr())
    plan = plans.get(key) if key is not None else None
    if plan is not None and plan.cap < cap:
        plan = None
    while True:
        if plan is None:
            plan = _AreaPlan(panel, nq, cap)
            if key is not None:
                # A kept plan pins its buffers (two hit buffers of `cap` 16-byte slots, the workspaces) and, from its second
                # use, a HIP graph: at most eight shapes and _PLAN_BYTES_MAX bytes per panel, oldest dropped first
                # (PackedPanel.clear_area_plans() drops them all).
                kept = [k for k in plans if k != "all_rows"]
                while kept and (len(kept) >= 8 or sum(plans[k].nbytes for k in kept) + plan.nbytes > _PLAN_BYTES_MAX):
                    del plans[kept.pop(0)]
                plans[key] = plan
        plan.uses += 1
        if key is not None and plan.graph is None and (use_graph or plan.uses >= 2) and \
                not torch.cuda.is_current_stream_capturing():
            try:                                    # capture the launches once; a failure leaves the eager path
                torch.cuda.current_stream().synchronize()   # this stream's launches are done (torch's graph context then
                                                            # synchronises the device once more, on every capture: its own rule)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):   # other threads (the shells' table workers) keep allocating
                    _area_launch(panel, pos, q, nq, flank, measure, thres, plan)
                plan.graph = g
            except Exception:                       # noqa: BLE001
                plan.graph = False
                torch.cuda.synchronize()
        if plan.graph:
            plan.graph.replay()
        else:
            _area_launch(panel, pos, q, nq, flank, measure, thres, plan, events)
        total, reserved = (int(x) for x in plan.summary.tolist())             # the one host round trip
        if reserved <= plan.cap:
            break
        cap = reserved + 4096            # the count is exact for a re-run: one retry suffices
        if key is not None:
            plans.pop(key, None)
        plan = None
    # the caller's copy of the result -- query / opposing rows as int64, the value pairs, and for a kept plan (whose buffers
    # the next call overwrites) the offsets index and the band's pass count -- in ONE launch (ldx_area_results_dev; it was
    # five small torch kernels, ~20 us of a 0.39 ms call)
    qrow = torch.empty(total, dtype=torch.int64, device=dev)
    orow = torch.empty(total, dtype=torch.int64, device=dev)
    ld32 = torch.empty((total, 2), dtype=torch.float32, device=dev)
    offsets = torch.empty_like(plan.offsets) if key is not None else plan.offsets
    use_band = get_area_path() != "popcount" and (get_area_path() != "auto" or (nq * 16 >= panel.n_snps and panel.n_snps >= 2))
    word = None
    if use_band:
        off = lib.ldx_area_band_passes_offset(panel.n_snps)
        word_src = plan.ws[off:off + 4]
        word = torch.empty(4, dtype=torch.uint8, device=dev) if key is not None else word_src
    check(lib.ldx_area_results_dev(plan.hits.data_ptr(), total, qrow.data_ptr(), orow.data_ptr(), ld32.data_ptr(),
                                   plan.offsets.data_ptr(), offsets.data_ptr() if key is not None else None,
                                   panel.n_snps + 1,
                                   word_src.data_ptr() if (use_band and key is not None) else None,
                                   word.data_ptr() if (use_band and key is not None) else None, _stream_ptr()),
          "ldx_area_results_dev")

    def count_pairs() -> int:
        # pairs evaluated = sum over queries of window population (bookkeeping, on device, only when asked for)
        qpos = pos[q.to(torch.int64)]
        lo = torch.searchsorted(pos, torch.clamp(qpos - flank, min=0), right=True)
        hi = torch.searchsorted(pos, qpos + flank, right=True)
        self_in = torch.clamp(qpos - flank, min=0) < qpos      # the query lies in its own window unless flank == 0
        return int((hi - lo).sum().item()) - int(self_in.sum().item())

    band = None
    if use_band:
        band = lambda: int(word.view(torch.int32).item())    # noqa: E731
    return AreaHits(qrow, orow, ld32, count_pairs, offsets, band)


# --------------------------------------------------------------------------- instrumentation
def probe_andpop(blocks: int, threads: int, iters: int) -> torch.Tensor:
    sink = torch.empty(blocks * threads, dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
    check(lib.ldx_probe_andpop_dev(sink.data_ptr(), blocks, threads, iters, _stream_ptr()), "ldx_probe_andpop_dev")
    return sink
