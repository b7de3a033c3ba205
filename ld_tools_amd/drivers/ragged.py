"""calc_ld for pairs of variants whose genotype lists differ in length (mixed ploidy: chrX across the PAR boundary).

The reference pairs such lists with zip (calc_ld.py:30-31): the haplotype count n and the alt/alt count cover the
first min(len_1, len_2) entries, while the allele counts cover each FULL list (calc_ld.py:37-40) -- so a frequency
may exceed 1 and the results follow whatever the formulas give.  The batched kernels need one haplotype count per
panel, so the variants are grouped by length; for every pair of groups the two code blocks are truncated to the
shorter length, packed, and one ``pair_counts`` launch gives the alt/alt counts of the whole block; the epilogue
(``ldx_ld_from_counts_ex_dev``: the op-for-op mirror with real divisions, exact for any counts) runs on the pairs
asked for.  Everything numeric happens on the GPU; the host only regroups indices.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from .. import _lib
from ..ops import pair_counts
from ..panel import PackedPanel, _stream_ptr, encode_codes


def ragged_pairs(genotype_rows: Sequence[Sequence], pairs: Sequence[Tuple[int, int]]) -> List[dict]:
    """``calc_ld(genotype_rows[i], genotype_rows[j])`` for every (i, j) of ``pairs`` (var_1 = i, var_2 = j): the
    reference's result dicts (Python floats rounded to 4 decimals, the int 0 in the degenerate branches)."""
    rows = [encode_codes(list(r)).ravel() for r in genotype_rows]
    if not pairs:
        return []
    used = sorted({k for p in pairs for k in p})
    if any(rows[k].size == 0 for k in used):
        raise ZeroDivisionError("division by zero")          # calc_ld.py:33 on an empty genotype list
    alt = {k: int((rows[k] == 1).sum()) for k in used}        # list.count over the FULL list (calc_ld.py:37-40)
    ref = {k: int((rows[k] == 0).sum()) for k in used}
    by_len: Dict[Tuple[int, int], List[int]] = {}
    for p, (i, j) in enumerate(pairs):
        by_len.setdefault((rows[i].size, rows[j].size), []).append(p)
    out: List[dict] = [None] * len(pairs)                     # type: ignore[list-item]
    dev = torch.device("cuda", torch.cuda.current_device())
    for (li, lj), members in by_len.items():
        n = min(li, lj)                                       # len(zip(...)) (calc_ld.py:30-31)
        ii = sorted({pairs[p][0] for p in members})
        jj = sorted({pairs[p][1] for p in members})
        pa = PackedPanel.from_codes(np.stack([rows[k][:n] for k in ii]).astype(np.int8, copy=False))
        pb = PackedPanel.from_codes(np.stack([rows[k][:n] for k in jj]).astype(np.int8, copy=False))
        block = pair_counts(pa, pb).cpu().numpy()             # n11 over the zipped prefix, |ii| x |jj|
        ri = {k: x for x, k in enumerate(ii)}
        rj = {k: x for x, k in enumerate(jj)}
        m = len(members)
        n11 = np.array([block[ri[pairs[p][0]], rj[pairs[p][1]]] for p in members], dtype=np.uint32)
        a1 = np.array([alt[pairs[p][0]] for p in members], dtype=np.uint32)
        r1 = np.array([ref[pairs[p][0]] for p in members], dtype=np.uint32)
        a2 = np.array([alt[pairs[p][1]] for p in members], dtype=np.uint32)
        r2 = np.array([ref[pairs[p][1]] for p in members], dtype=np.uint32)
        t = [torch.from_numpy(x.view(np.int32).copy()).to(dev) for x in (n11, a1, r1, a2, r2)]
        k = torch.empty((m, 2), dtype=torch.float64, device=dev)
        raw = torch.empty((m, 2), dtype=torch.float64, device=dev)
        flags = torch.empty(m, dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib.ldx_ld_from_counts_ex_dev(int(n), m, *(x.data_ptr() for x in t), raw.data_ptr(), k.data_ptr(),
                                                      None, None, flags.data_ptr(), _stream_ptr()),
                   "ldx_ld_from_counts_ex_dev")
        kk = k.cpu().numpy()
        ff = flags.cpu().numpy() & 3
        for x, p in enumerate(members):
            i, j = pairs[p]
            out[p] = {"r_square": 0 if ff[x] & _lib.FLAG_RSQ_INT0 else float(kk[x, 0]) / 10000.0,
                      "d_prime": 0 if ff[x] & _lib.FLAG_DPRIME_INT0 else float(kk[x, 1]) / 10000.0,
                      "var_1_alt_freq": round(alt[i] / n, 4),      # calc_ld.py:41-44,96-97: full-list count / zip length
                      "var_2_alt_freq": round(alt[j] / n, 4)}
    return out
