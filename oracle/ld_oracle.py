"""CPU oracle for the pairwise-LD hot path -- TEST INFRASTRUCTURE ONLY.

This module is a from-scratch CPU restatement of what PlatonB/ld-tools computes in
``backend/calc_ld.py:3-99`` and of the two loops that drive it
(``ld_triangle.py:133-230`` and ``ld_area.py:152-276``).  It exists so that the
HIP path can be checked against something that travels to the GPU box (the
reference's own file cannot).  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``ld_tools_amd`` never does.

Parity pin: every function below is checked against the real reference (imported
in the build container from /root/reference) by ``tests/golden/make_golden.py``;
the resulting vectors are committed under ``tests/golden/`` and re-checked by
``tests/test_oracle_golden.py`` on every run.

Two levels are provided:

* ``calc_ld_lists``     -- per-haplotype restatement on Python sequences
                           (what the reference does, scan by scan).
* ``ld_from_counts``    -- the same result as a pure function of the six integers
                           (n, n11, a1, r1, a2, r2); this is the form the GPU
                           epilogue mirrors op for op in fp64.
"""
from __future__ import annotations

import math

import numpy as np

KEYS = ("r_square", "d_prime", "var_1_alt_freq", "var_2_alt_freq")

# flag bits shared with include/ldx.h (LDX_FLAG_*)
FLAG_DPRIME_INT0 = 1   # d_prime is the *int* 0 of calc_ld.py:68-69 / :75-76
FLAG_RSQ_INT0 = 2      # r_square is the *int* 0 of calc_ld.py:89-90


# --------------------------------------------------------------------------- #
# scalar restatement
# --------------------------------------------------------------------------- #
def pair_counts_lists(g1, g2):
    """The six integers calc_ld derives from two allele-code sequences.

    calc_ld.py:30-32  n   = number of zipped haplotypes (shorter sequence wins)
                      n11 = haplotypes carrying code 1 at both variants
    calc_ld.py:37-40  a/r = per-variant counts of code 1 / code 0 over the FULL
                      sequence (not the zipped prefix); any other code (None,
                      2, ...) is in n but in neither count.
    """
    g1 = g1 if isinstance(g1, (list, tuple)) else list(g1)
    g2 = g2 if isinstance(g2, (list, tuple)) else list(g2)
    pairs = list(zip(g1, g2))            # as many haplotypes as the shorter sequence
    n = len(pairs)
    n11 = pairs.count((1, 1))            # tuple equality: x == 1 and y == 1
    return n, n11, g1.count(1), g1.count(0), g2.count(1), g2.count(0)


def ld_raw_from_counts(n, n11, a1, r1, a2, r2):
    """Unrounded (r_square, d_prime, fa1, fa2, flags) in the reference's op order.

    calc_ld.py:33,41-44  five true divisions by n (ZeroDivisionError if n == 0)
    calc_ld.py:50        d = f11 - fa1*fa2          (product rounds, then the subtraction)
    calc_ld.py:63-76     sign-branched D' with ZeroDivisionError -> int 0
    calc_ld.py:86-90     r2 = d**2 / (((fa1*fr1)*fa2)*fr2) gated on unrounded D' != 0
    """
    f11 = n11 / n
    fa1 = a1 / n
    fr1 = r1 / n
    fa2 = a2 / n
    fr2 = r2 / n
    d = f11 - fa1 * fa2
    flags = 0
    if d >= 0:
        bound = min(fa1 * fr2, fr1 * fa2)
    else:
        bound = max(-fa1 * fa2, -fr1 * fr2)
    if bound == 0:            # float division by +-0.0 raises in Python
        d_prime = 0
        flags |= FLAG_DPRIME_INT0
    else:
        d_prime = d / bound
    if d_prime != 0:
        r_square = (d ** 2) / (fa1 * fr1 * fa2 * fr2)
    else:
        r_square = 0
        flags |= FLAG_RSQ_INT0
    return r_square, d_prime, fa1, fa2, flags


def ld_from_counts(n, n11, a1, r1, a2, r2):
    """Result dict of calc_ld (calc_ld.py:94-99) from the six integers."""
    r_square, d_prime, fa1, fa2, _ = ld_raw_from_counts(n, n11, a1, r1, a2, r2)
    return {
        "r_square": round(r_square, 4),
        "d_prime": round(d_prime, 4),
        "var_1_alt_freq": round(fa1, 4),
        "var_2_alt_freq": round(fa2, 4),
    }


def calc_ld_lists(g1, g2):
    """Drop-in behaviour of calc_ld(var_1_genotypes, var_2_genotypes)."""
    return ld_from_counts(*pair_counts_lists(g1, g2))


def triangle_rows_lists(rows, i0: int, i1: int) -> int:
    """calc_ld_lists for every pair (i, j), i0 <= i < i1, j < i -- one worker's share when the pair loop is spread
    over a process pool the way the reference spreads input files (ld_triangle.py:394-409).  Returns the pair count."""
    pairs = 0
    for i in range(i0, i1):
        gi = rows[i]
        for j in range(i):
            calc_ld_lists(gi, rows[j])
            pairs += 1
    return pairs


def round4(x: float) -> float:
    """Exact emulation of Python's round(x, 4) for finite x >= 0, float ops only.

    calc_ld.py:94-97 rounds with the builtin, which rounds the *exact* binary
    value to 4 decimals, ties to even.  1e4 is exact in binary64, so
    x*1e4 == y + e exactly with y = fl(x*1e4), e = fma(x, 1e4, -y); a
    half-integer lying strictly between y and y+e would itself be a double
    closer to the exact product than y, so deciding on frac(y) and, only when
    frac(y) == 0.5, on the sign of e reproduces the correctly rounded
    decision.  k/1e4 is one correctly rounded division == the double nearest
    to the decimal string Python re-parses.  This is the form the C oracle and
    the HIP epilogue use; it is checked against round() in the tests.
    """
    y = x * 1e4
    e = math.fma(x, 1e4, -y) if hasattr(math, "fma") else _fma_residual(x, y)
    k = math.floor(y)
    f = y - k
    if f > 0.5:
        k += 1
    elif f == 0.5:
        if e > 0 or (e == 0 and (int(k) & 1)):
            k += 1
    return k / 1e4


def _fma_residual(x: float, y: float) -> float:
    """x*1e4 - y computed exactly (Python 3.10 has no math.fma): rationals."""
    from fractions import Fraction
    return float(Fraction(x) * 10000 - Fraction(y))


# --------------------------------------------------------------------------- #
# genotype codes and bit-planes (numpy, used for panels)
# --------------------------------------------------------------------------- #
def encode_codes(seq) -> np.ndarray:
    """Allele codes of one variant as int8: 1 = ALT, 0 = REF, 2 = anything else.

    Membership is by ``== 1`` / ``== 0`` exactly as list.count does it
    (calc_ld.py:37-40): 1.0 and True count as 1, None / 2 / '1' count as neither.
    """
    out = np.empty(len(seq), dtype=np.int8)
    for k, v in enumerate(seq):
        out[k] = 1 if v == 1 else (0 if v == 0 else 2)
    return out


def pack_planes(codes: np.ndarray):
    """[N][H] int8 codes -> (alt, ref) bit-planes, row-major [N][W64] uint64.

    Bit h of row i of ``alt`` is set iff codes[i][h] == 1, of ``ref`` iff == 0;
    pad bits are zero.  (Plain row-major here; the device library uses its own
    tiled layout and exposes converters.)
    """
    codes = np.asarray(codes, dtype=np.int8)
    n, h = codes.shape
    w64 = (h + 63) // 64
    def plane(mask):
        padded = np.zeros((n, w64 * 64), dtype=np.uint8)
        padded[:, :h] = mask
        b = np.packbits(padded, axis=1, bitorder="little")
        return b.view(np.uint64).reshape(n, w64)
    return plane(codes == 1), plane(codes == 0)


_POP8 = np.array([bin(i).count("1") for i in range(256)], dtype=np.uint32)


def popcount_rows(plane: np.ndarray) -> np.ndarray:
    return _POP8[plane.view(np.uint8)].reshape(plane.shape[0], -1).sum(axis=1).astype(np.uint32)


def pair_n11(alt: np.ndarray, rows: np.ndarray, cols: np.ndarray) -> np.ndarray:
    """n11[len(rows)][len(cols)] = popcount(alt[row] & alt[col])  (calc_ld.py:32)."""
    out = np.empty((len(rows), len(cols)), dtype=np.uint32)
    a8 = alt.view(np.uint8).reshape(alt.shape[0], -1)
    for k, i in enumerate(rows):
        out[k] = _POP8[a8[i][None, :] & a8[cols]].sum(axis=1)
    return out


# --------------------------------------------------------------------------- #
# vectorised epilogue (numpy float64, same op order; d*d instead of pow(d, 2))
# --------------------------------------------------------------------------- #
def ld_raw_from_counts_np(n, n11, a1, r1, a2, r2):
    """Array form of ld_raw_from_counts.  Returns (r2, dprime, flags) unrounded.

    Uses d*d for the square: numpy's pow loop is not guaranteed to be the libm
    pow CPython calls for ``d ** 2`` (calc_ld.py:87).  The scalar functions above
    are the pinned ones; this helper is for large-panel cross-checks at 1e-12.
    """
    n11 = np.asarray(n11, dtype=np.float64)
    nn = np.float64(n)
    with np.errstate(divide="ignore", invalid="ignore"):
        f11 = n11 / nn
        fa1 = np.asarray(a1, dtype=np.float64) / nn
        fr1 = np.asarray(r1, dtype=np.float64) / nn
        fa2 = np.asarray(a2, dtype=np.float64) / nn
        fr2 = np.asarray(r2, dtype=np.float64) / nn
        p = fa1 * fa2
        d = f11 - p
        dmax = np.minimum(fa1 * fr2, fr1 * fa2)
        dmin = np.maximum(-p, -(fr1 * fr2))
        bound = np.where(d >= 0, dmax, dmin)
        zero_bound = bound == 0
        dp = np.where(zero_bound, 0.0, d / np.where(zero_bound, 1.0, bound))
        den = ((fa1 * fr1) * fa2) * fr2
        gate = dp != 0
        rsq = np.where(gate, (d * d) / np.where(gate, den, 1.0), 0.0)
    flags = (zero_bound.astype(np.uint8) * FLAG_DPRIME_INT0) | ((~gate).astype(np.uint8) * FLAG_RSQ_INT0)
    return rsq, dp, flags


def round4_np(x: np.ndarray) -> np.ndarray:
    """Array form of round4 via Python's round (exact, slow-ish; test sizes only)."""
    flat = np.asarray(x, dtype=np.float64).ravel()
    return np.array([round(float(v), 4) for v in flat], dtype=np.float64).reshape(np.shape(x))


# --------------------------------------------------------------------------- #
# drivers (restated from the reference's loops, composed from the scalar oracle)
# --------------------------------------------------------------------------- #
def triangle_lists(genotype_rows, measure="r_square", thres=None, calc=calc_ld_lists):
    """ld_two_dim of ld_triangle.py:114,133-230 for position-sorted variants.

    Lower triangle (row > col) holds calc(row, col)[measure]; cells whose rounded
    measure is below ``thres`` keep the int 0 they were initialised with, as do the
    diagonal and the upper triangle.
    """
    n = len(genotype_rows)
    m = [[0 for _ in range(n)] for _ in range(n)]
    for row in range(n):
        for col in range(row):
            vals = calc(genotype_rows[row], genotype_rows[col])
            if thres is not None and vals[measure] < thres:
                continue
            m[row][col] = vals[measure]
    return m


def area_lists(genotype_rows, positions, query_idx, flank, measure="r_square", thres=0.8,
               calc=calc_ld_lists):
    """Hit lists of ld_area.py:174-177,215-276 over an in-memory panel.

    ``positions`` are ascending (VCF order).  For every query index q the window is
    [max(0, pos_q - flank), pos_q + flank] in pysam's half-open 0-based fetch
    coordinates, i.e. 1-based positions p with low < p <= high; the query itself is
    skipped; a hit is (q, o, var_2_alt_freq, r_square, d_prime, pos_o - pos_q) kept
    when the rounded ``measure`` >= thres, in VCF order.
    """
    hits = []
    for q in query_idx:
        low = max(0, positions[q] - flank)
        high = positions[q] + flank
        for o in range(len(genotype_rows)):
            if o == q or not (low < positions[o] <= high):
                continue
            vals = calc(genotype_rows[q], genotype_rows[o])
            if vals[measure] < thres:
                continue
            hits.append((q, o, vals["var_2_alt_freq"], vals["r_square"], vals["d_prime"],
                         positions[o] - positions[q]))
    return hits
