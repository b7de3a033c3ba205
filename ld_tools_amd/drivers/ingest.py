"""Genotype ingest: VariantRecord-like objects -> allele-code rows."""
from __future__ import annotations

from typing import Iterable, List, Sequence

import numpy as np

from ..panel import encode_codes


class RaggedGenotypesError(ValueError):
    """Variants of one panel carry different haplotype counts (mixed ploidy, e.g. chrX across the PAR boundary).

    The reference pairs such lists with zip (n = the shorter one, calc_ld.py:30-31) while counting alleles over the
    full lists; one packed panel needs one haplotype count.  ``codes_matrix`` raises this; the triangle and area
    drivers catch it and go through ``drivers/ragged.py`` (the same semantics, batched per pair of lengths).
    """


def sample_genotypes(rec, sample_names: Sequence[str]) -> list:
    """The reference's genotype list of one variant (ld_triangle.py:167-171, ld_area.py:183-187,231-235,
    ld_lite.py:119-123): the GT tuples of the selected samples concatenated in ``sample_names`` order, samples
    the record does not carry skipped (KeyError).  Haplotype h of sample s sits at index 2*s + phase for
    diploid calls; codes are whatever the record holds (0, 1, 2, None, ...)."""
    genotypes: list = []
    samples = rec.samples
    for name in sample_names:
        try:
            genotypes += samples[name]["GT"]
        except KeyError:
            continue
    return genotypes


def find_record(vcf, chrom, pos: int, rs_id: str):
    """The record the reference works with for [pos, rs_id] (ld_triangle.py:160-165, ld_area.py:153-159):
    the first record of fetch(chrom, pos - 1, pos) whose id equals rs_id; None if there is none."""
    for rec in vcf.fetch(chrom, pos - 1, pos):
        if rec.id == rs_id:
            return rec
    return None


def codes_matrix(genotype_rows: Iterable[Sequence]) -> np.ndarray:
    """int8 [n_variants][n_haplotypes] code matrix for PackedPanel.from_codes (1 = ALT, 0 = REF, 2 = neither)."""
    rows: List[np.ndarray] = [encode_codes(list(r)) for r in genotype_rows]
    if not rows:
        raise ValueError("no variants")
    width = rows[0].size
    if any(r.size == 0 for r in rows):
        raise ZeroDivisionError("division by zero")   # calc_ld.py:33 on an empty genotype list (no matching record)
    if any(r.size != width for r in rows):
        raise RaggedGenotypesError(f"haplotype counts differ between variants: {sorted({int(r.size) for r in rows})}")
    return np.stack(rows).astype(np.int8, copy=False)


def k_to_python(values32: np.ndarray, fixes=None):
    """Device float32 results -> the reference's Python values: float ``k / 10**4`` (== round(x, 4)), or the int 0
    where the value carries the int-0 mark (-0.0f; calc_ld.py:68-69,75-76,89-90).  ``fixes`` maps the index of an
    escape cell (NaN: a value the float32 cannot identify, >= 1024) -- flat, or (row, col) for a 2-D input -- to its exact
    Python value (TriangleResult.dense_values / ops.ld_pairs); an escape without a fix is an error, never a guess."""
    v = np.asarray(values32, dtype=np.float32)
    esc = np.isnan(v)
    k = np.rint(np.where(esc, 0, v).astype(np.float64) * 1e4)
    int0 = np.signbit(v) & (v == 0)
    flat_k, flat_z = k.ravel(), int0.ravel()
    out = [0 if z else kk / 10000.0 for kk, z in zip(flat_k.tolist(), flat_z.tolist())]
    if esc.any():
        fixes = fixes or {}
        for flat in np.flatnonzero(esc.ravel()).tolist():
            key = flat if v.ndim < 2 else tuple(int(x) for x in np.unravel_index(flat, v.shape))
            if key not in fixes:
                raise ValueError(f"escape cell {key} has no exact value: resolve it with ops.ld_pairs")
            out[flat] = fixes[key]
    return out
