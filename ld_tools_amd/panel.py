"""PackedPanel: the 2-bit/allele SNP x haplotype matrix resident in HBM.

Replaces the per-pair genotype list assembly of the reference (ld_triangle.py:160-186,
ld_area.py:182-187,230-235): every SNP is turned into an ALT bit-row and a REF bit-row once,
in the tiled layout described in include/ldx.h, together with its allele counts
(calc_ld.py:37-40) and frequency vectors (calc_ld.py:41-44).

torch is used only as the owner of device memory and of the HIP stream; all arithmetic runs
in libldx.so.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.LdxError("ld_tools_amd needs a HIP device (MI355X / gfx950); there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def encode_codes(genotypes) -> np.ndarray:
    """Allele codes as int8 with the membership rule of list.count (calc_ld.py:37-40).

    1 (also 1.0, True) -> 1 = ALT; 0 (0.0, False) -> 0 = REF; anything else (None for a missing
    GT, 2 for a second ALT, strings, NaN) -> 2 = in neither count.  Accepts a sequence (one
    variant) or a 2-D array-like (variants x haplotypes); numeric numpy input takes a
    vectorised path, object input is compared element by element.
    """
    if isinstance(genotypes, (list, tuple)):
        # the common case -- a flat list of small non-negative ints (0 / 1, the odd 2) -- goes through bytes(): ten times
        # faster than numpy's element-by-element conversion; None, floats, nested rows or large values raise and take the
        # general path below
        try:
            out = np.frombuffer(bytes(genotypes), dtype=np.uint8).copy()
            out[out > 1] = 2
            return out.view(np.int8)
        except (TypeError, ValueError):
            pass
        try:                                # still numeric (floats, large or negative ints, nested rows): vectorised
            arr = np.asarray(genotypes)
            if arr.dtype.kind not in "biuf":
                raise TypeError
        except (TypeError, ValueError):    # None, strings, mixed or ragged content: compared element by element as objects
            arr = np.empty(len(genotypes), dtype=object)
            arr[:] = list(genotypes)
            if all(isinstance(v, (list, tuple)) for v in genotypes) and len({len(v) for v in genotypes}) == 1 and genotypes:
                arr = np.array([list(v) for v in genotypes], dtype=object)
    else:
        arr = genotypes if isinstance(genotypes, np.ndarray) else np.asarray(genotypes, dtype=None)
    if arr.dtype == object or arr.dtype.kind in "USV":
        flat = arr.ravel()
        out = np.fromiter((1 if v == 1 else (0 if v == 0 else 2) for v in flat), dtype=np.int8,
                          count=flat.size)
        return out.reshape(arr.shape)
    out = np.full(arr.shape, 2, dtype=np.int8)
    out[arr == 1] = 1
    out[arr == 0] = 0
    return out


@dataclass
class PackedPanel:
    """Device-resident packed genotype panel (one chromosome / one sample selection)."""

    n_snps: int
    n_hap: int
    alt: torch.Tensor          # uint8 [plane_bytes]   tiled ALT plane
    ref: torch.Tensor          # uint8 [plane_bytes]   tiled REF plane
    acnt: torch.Tensor         # int32 [padded_snps]   count of code 1 per SNP (bit pattern of uint32)
    rcnt: torch.Tensor         # int32 [padded_snps]   count of code 0 per SNP
    fa: torch.Tensor           # float64 [padded_snps] a / n
    fr: torch.Tensor           # float64 [padded_snps] r / n
    q: torch.Tensor            # float64 [padded_snps] fa * fr

    # ------------------------------------------------------------------ construction
    @staticmethod
    def empty(n_snps: int, n_hap: int, device: Optional[torch.device] = None) -> "PackedPanel":
        dev = device or require_gpu()
        if not (1 <= n_hap <= _lib.MAX_HAPS):
            raise _lib.LdxError(f"n_hap={n_hap} outside 1..{_lib.MAX_HAPS} (LDX_MAX_HAPS)")
        if n_snps < 1:
            raise _lib.LdxError("a panel needs at least one SNP")
        pb = lib.ldx_plane_bytes(n_snps, n_hap)
        npad = lib.ldx_padded_snps(n_snps)
        z = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)  # noqa: E731
        return PackedPanel(n_snps, n_hap, z(pb, torch.uint8), z(pb, torch.uint8), z(npad, torch.int32),
                           z(npad, torch.int32), z(npad, torch.float64), z(npad, torch.float64),
                           z(npad, torch.float64))

    @staticmethod
    def from_codes(codes, device: Optional[torch.device] = None) -> "PackedPanel":
        """Pack an int8 [n_snps][n_hap] code matrix (numpy or torch, host or device)."""
        dev = device or require_gpu()
        if isinstance(codes, np.ndarray):
            codes = torch.from_numpy(np.ascontiguousarray(codes, dtype=np.int8))
        if codes.dtype != torch.int8 or codes.dim() != 2:
            raise _lib.LdxError("codes must be an int8 matrix [n_snps][n_hap]")
        codes = codes.to(dev)
        if codes.stride(1) != 1:
            codes = codes.contiguous()      # a row-strided view (padded leading dimension) is taken as is
        n_snps, n_hap = codes.shape
        p = PackedPanel.empty(n_snps, n_hap, dev)
        p.pack_from(codes)
        return p

    @staticmethod
    def from_genotypes(rows, device: Optional[torch.device] = None) -> "PackedPanel":
        """Pack per-variant genotype sequences as calc_ld receives them (lists of 0/1/None/...)."""
        rows = list(rows)
        width = max(len(r) for r in rows)
        codes = np.full((len(rows), width), 2, dtype=np.int8)
        for k, r in enumerate(rows):
            codes[k, :len(r)] = encode_codes(list(r))
        return PackedPanel.from_codes(codes, device)

    def pack_from(self, codes: torch.Tensor) -> None:
        """(Re)pack this panel from a device int8 matrix of its shape, on the current stream."""
        s = _stream_ptr()
        check(lib.ldx_pack_codes_dev(codes.data_ptr(), self.n_snps, self.n_hap, codes.stride(0),
                                     self.alt.data_ptr(), self.ref.data_ptr(), self.acnt.data_ptr(),
                                     self.rcnt.data_ptr(), s), "ldx_pack_codes_dev")
        self.refresh_stats()

    def refresh_stats(self) -> None:
        check(lib.ldx_snp_stats_dev(self.acnt.data_ptr(), self.rcnt.data_ptr(), self.n_snps, self.n_hap,
                                    self.fa.data_ptr(), self.fr.data_ptr(), self.q.data_ptr(), _stream_ptr()),
              "ldx_snp_stats_dev")

    # ------------------------------------------------------------------ geometry
    @property
    def padded_snps(self) -> int:
        return lib.ldx_padded_snps(self.n_snps)

    @property
    def n_pairs(self) -> int:
        return self.n_snps * (self.n_snps - 1) // 2

    @property
    def n_units(self) -> int:
        return lib.ldx_triangle_units(self.n_snps)

    @property
    def device(self) -> torch.device:
        return self.alt.device

    # ------------------------------------------------------------------ per-SNP results
    def alt_counts(self) -> np.ndarray:
        return self.acnt[: self.n_snps].cpu().numpy().view(np.uint32)

    def ref_counts(self) -> np.ndarray:
        return self.rcnt[: self.n_snps].cpu().numpy().view(np.uint32)

    def alt_freq4(self) -> torch.Tensor:
        """round(a/n, 4) per SNP (calc_ld.py:96-97, ld_area.py:188-189), float64 on device."""
        out = torch.empty(self.n_snps, dtype=torch.float64, device=self.device)
        check(lib.ldx_alt_freq4_dev(self.acnt.data_ptr(), self.n_snps, self.n_hap, out.data_ptr(),
                                    _stream_ptr()), "ldx_alt_freq4_dev")
        return out

    def clear_area_plans(self) -> None:
        """Drop the ld_area plans kept on this panel (ops.ld_area: buffers + HIP graph per call shape) and their memory."""
        self.__dict__.pop("_area_plans", None)
