"""Drop-in replacement for the reference's ``backend/calc_ld.py``.

Same module path tail, same function name, same signature and the same return value:

    from ld_tools_amd.backend.calc_ld import calc_ld
    calc_ld(var_1_genotypes, var_2_genotypes)
        -> {'r_square': ..., 'd_prime': ..., 'var_1_alt_freq': ..., 'var_2_alt_freq': ...}

Behaviour preserved from calc_ld.py:3-99 (each checked in tests/ against golden vectors made
with the reference itself):
  * values are Python floats rounded to 4 decimals, or the *int* 0 in the degenerate branches
    (calc_ld.py:68-69,75-76,89-90), so ``str()`` of a result is identical;
  * allele codes are matched with ``== 1`` / ``== 0`` as ``list.count`` does: ``None``, ``2`` ...
    are haplotypes that belong to neither allele count (calc_ld.py:37-40);
  * unequal lengths: the haplotype count is the shorter length (zip, calc_ld.py:30-31) while
    allele counts cover each full sequence;
  * empty input raises ZeroDivisionError (calc_ld.py:33).
Extension: numpy arrays are accepted as well as lists/tuples.

The arithmetic -- packing, the alt/alt haplotype count, D, D', r^2 and the 4-decimal rounding --
runs on the GPU through ``ldx_calc_ld_host`` (include/ldx.h).  There is no CPU implementation
in this module; without the HIP library the import fails.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib
from ..panel import encode_codes

__version__ = "V5.1-ldx"


class _Ld64(C.Structure):
    _fields_ = [("r_square", C.c_double), ("d_prime", C.c_double)]


def calc_ld_full(var_1_genotypes, var_2_genotypes):
    """calc_ld plus the intermediate integers: (result dict, counts, raw, flags).

    counts = (n, n11, a1, r1, a2, r2); raw = (r_square, d_prime) unrounded.
    """
    g1 = np.ascontiguousarray(encode_codes(var_1_genotypes).ravel())
    g2 = np.ascontiguousarray(encode_codes(var_2_genotypes).ravel())
    if g1.size == 0 or g2.size == 0:
        raise ZeroDivisionError("division by zero")      # calc_ld.py:33 with an empty zip
    counts = (C.c_uint32 * 6)()
    raw, rnd = _Ld64(), _Ld64()
    freq = (C.c_double * 2)()
    flags = C.c_uint8(0)
    _lib.check(_lib.lib.ldx_calc_ld_host(g1.ctypes.data, g1.size, g2.ctypes.data, g2.size, counts,
                                         C.byref(raw), C.byref(rnd), freq, C.byref(flags)),
               "ldx_calc_ld_host")
    f = flags.value
    result = {
        "r_square": 0 if f & _lib.FLAG_RSQ_INT0 else rnd.r_square,
        "d_prime": 0 if f & _lib.FLAG_DPRIME_INT0 else rnd.d_prime,
        "var_1_alt_freq": freq[0],
        "var_2_alt_freq": freq[1],
    }
    return result, tuple(int(c) for c in counts), (raw.r_square, raw.d_prime), f


def calc_ld(var_1_genotypes, var_2_genotypes):
    """LD (r2, D') of two biallelic variants from their phased allele codes.

    Arguments are two sequences of numeric allele codes in "single" form (``1|0`` -> ``1, 0``),
    one entry per haplotype of the selected samples, 0 = reference allele, 1 = alternative.
    """
    return calc_ld_full(var_1_genotypes, var_2_genotypes)[0]
