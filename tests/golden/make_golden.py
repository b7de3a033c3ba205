#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py [--out DIR]        (default: tests/golden/ itself)

It loads the reference's ``backend/calc_ld.py`` BY FILE PATH (tests/golden/_reference.py: this repo's own
top-level ``backend/`` package would shadow it on sys.path; pure Python, no imports of its own) and
records its outputs on inputs produced by this repo's own deterministic generator
(ld_tools_amd/synth.py) or realised from count tuples.  Only inputs (or their seeds) and
expected outputs are stored -- no reference source text.

Files written
  kat_counts.json      F1: known-answer count tuples + literal-list probes
  small_n.npz          F2: every reachable (n, n11, a1, r1, a2, r2) for n in {4, 8} (all 3x3 joint
                            tables) and every a+r == n tuple for n in {16, 37, 100}
  panels.npz           F3: all-pairs results on five synthetic panels (incl. code-2 haplotypes)
  drivers.json         F4: ld_triangle matrices and ld_area hit lists composed from reference calls
"""
from __future__ import annotations

import hashlib
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
OUT = HERE                       # --out DIR redirects every file written (the regen-and-compare test)
sys.dont_write_bytecode = True
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT))

import _reference  # noqa: E402

ref_calc_ld = _reference.load_reference_calc_ld()      # the reference, checked by code-object file name

from ld_tools_amd import synth  # noqa: E402
from oracle import ld_oracle as orc  # noqa: E402


# ------------------------------------------------------------------------------------------
def realise(n, n11, a1, r1, a2, r2):
    """Two code lists (codes 0, 1, 2) with exactly these counts; None if infeasible."""
    o1, o2 = n - a1 - r1, n - a2 - r2
    if min(o1, o2, n11, a1 - n11, a2 - n11) < 0:
        return None
    # rows: var_1 code (1, 0, 2); columns: var_2 code (1, 0, 2); cell[1][1] fixed = n11
    row = {1: a1 - n11, 0: r1, 2: o1}
    col = {1: a2 - n11, 0: r2, 2: o2}
    cells = {(1, 1): n11}
    for (rc, cc) in [(1, 0), (1, 2), (0, 1), (2, 1), (0, 0), (0, 2), (2, 0), (2, 2)]:
        take = min(row[rc], col[cc])
        cells[(rc, cc)] = take
        row[rc] -= take
        col[cc] -= take
    if any(row.values()) or any(col.values()):
        return None
    g1, g2 = [], []
    for (rc, cc), k in cells.items():
        g1 += [rc] * k
        g2 += [cc] * k
    assert len(g1) == n
    assert orc.pair_counts_lists(g1, g2) == (n, n11, a1, r1, a2, r2)
    return g1, g2


def enc(v):
    """(k, is_int0): k = value * 10^4 as an exact integer; is_int0 = the value is the int 0."""
    if isinstance(v, int) and not isinstance(v, bool):
        assert v == 0
        return 0, 1
    k = round(v * 1e4)
    assert k / 1e4 == v, (v, k)
    return k, 0


def ref_counts(n, n11, a1, r1, a2, r2):
    lists = realise(n, n11, a1, r1, a2, r2)
    assert lists is not None, (n, n11, a1, r1, a2, r2)
    return ref_calc_ld(*lists)


# ------------------------------------------------------------------------------------------
KAT = [
    (5008, 1200, 2504, 2504, 2504, 2504), (5008, 2504, 2504, 2504, 2504, 2504), (5008, 0, 2504, 2504, 2504, 2504),
    (5008, 1252, 2504, 2504, 2504, 2504), (5008, 1, 1, 5007, 1, 5007), (5008, 0, 1, 5007, 1, 5007),
    (5008, 1, 1, 5007, 5007, 1), (5008, 3000, 3694, 1314, 3288, 1720), (5008, 2937, 3694, 1314, 2937, 2071),
    (5008, 17, 50, 4958, 1700, 3308), (5008, 0, 0, 5008, 2504, 2504), (5008, 2504, 5008, 0, 2504, 2504),
    (5008, 5008, 5008, 0, 5008, 0), (5008, 313, 626, 4382, 2504, 2504), (5008, 400, 626, 4382, 2504, 2504),
    (5008, 100, 626, 4382, 2504, 2504), (5008, 1000, 2400, 2500, 2300, 2600), (5008, 900, 2400, 2500, 2300, 2600),
    (1008, 331, 662, 346, 590, 418), (1008, 400, 662, 346, 590, 418), (1008, 252, 504, 504, 504, 504),
    (1008, 3, 5, 1003, 700, 308), (100, 19, 68, 32, 20, 80), (100, 30, 32, 68, 92, 8), (32, 1, 8, 24, 4, 28),
    (16, 3, 8, 8, 4, 12), (8, 2, 4, 4, 4, 4),
    # a + r < n with a tiny bound: D' far above 1
    (5008, 40, 40, 4000, 60, 1), (1008, 10, 10, 900, 500, 2), (100, 5, 5, 90, 50, 1),
]

LITERAL = [
    ([1, 0, 1, 0, 1, 1, 0, 0], [1, 0, 1, 0, 0, 1, 0, 1]),
    ([1, 1, 1, 1, 0, 0, 0, 0], [0, 0, 0, 0, 1, 1, 1, 1]),
    ([1, 0, None, 2, 1, 1, 0, 0], [1, 0, 1, 0, None, 1, 0, 2]),
    ([1.0, 0.0, 1.0, 0.0], [1.0, 1.0, 0.0, 0.0]),
    ([True, False, True, True], [1, 0, 0, 1]),
    ([1, 0, 1, 0, 1, 1], [1, 0, 1, 0]),            # unequal lengths: zip truncates, counts do not
    ([1, 0, 1], [1, 1, 0, 0, 1, 0, 1]),
    ([1, 1, 1, 1], [1, 0, 1, 0]),                  # monomorphic var_1
    ([0, 0, 0, 0], [0, 0, 0, 0]),
    ([2, None, "1", 1, 0], [1, 1, 1, 0, 0]),
    ((1, 0, 0, 1, 1, 0), (0, 1, 1, 0, 0, 1)),      # tuples
]


# The reference's only real-data known answers: values it PUBLISHES, inverted here to count tuples by
# running the reference itself over every feasible (a1, a2, n11) with a + r == n.  `n` = haplotypes of the panel the value
# was made on; the generator asserts that exactly one tuple reproduces all four published numbers.
#   README.md:168-193      plotly Figure dump, EUR panel (503 samples -> n = 1006), chr6: rs1521 / rs8084 / rs7192
#   gallery/ld_lite_tabular_output.png   rs10134555 / rs11624464, chr14: r2 0.7807, D' 0.9144, alt_freq 0.5247 / 0.5418.
#       The panel behind the screenshot is not stated and is NOT the ALL panel: no a / 5008 rounds to 0.5247.  Every even
#       n <= 5008 was searched once (oracle, then confirmed with the reference here): exactly five n have a solution, each a
#       unique one; all five are kept as known answers of the function.
PUBLISHED = [
    # (source, n, var_1_alt_freq, var_2_alt_freq, r_square, d_prime)   var_1 = the row (larger position), ld_triangle.py:193
    ("README.md:168-193 rs8084 x rs1521 (EUR)", 1006, 0.5865, 0.7376, 0.0003, 0.0247),
    ("README.md:168-193 rs7192 x rs1521 (EUR)", 1006, 0.6332, 0.7376, 0.0027, 0.0668),
    ("README.md:168-193 rs7192 x rs8084 (EUR)", 1006, 0.6332, 0.5865, 0.8216, 1.0),
] + [(f"gallery/ld_lite_tabular_output.png rs10134555 x rs11624464 (panel unknown; n = {n})", n, 0.5247, 0.5418, 0.7807, 0.9144)
     for n in (1700, 3160, 3400, 4568, 4860)]
NO_SOLUTION = [("gallery/ld_lite_tabular_output.png at the ALL panel", 5008, 0.5247, 0.5418, 0.7807, 0.9144)]


def invert_published(n, f1, f2, rsq, dp):
    """Every (n, n11, a1, r1, a2, r2), a + r == n, for which the REFERENCE returns exactly the four published values."""
    want = {"r_square": rsq, "d_prime": dp, "var_1_alt_freq": f1, "var_2_alt_freq": f2}
    sols = []
    for a1 in (a for a in range(n + 1) if round(a / n, 4) == f1):        # calc_ld.py:41,96
        for a2 in (a for a in range(n + 1) if round(a / n, 4) == f2):
            # the oracle proposes, the reference disposes: only n11 whose oracle result matches are replayed on lists
            for n11 in range(max(0, a1 + a2 - n), min(a1, a2) + 1):
                t = (n, n11, a1, n - a1, a2, n - a2)
                if orc.ld_from_counts(*t) == want:
                    sols.append(t)
            lo, hi = max(0, a1 + a2 - n), min(a1, a2)
            # ... and the reference is also run on the neighbours of every proposal and on both ends of the range, so a
            # disagreement between oracle and reference next to a solution would show as a second / missing solution
            check = {lo, hi} | {m for s in sols for m in (s[1] - 1, s[1], s[1] + 1) if lo <= m <= hi}
            got = [(n, m, a1, n - a1, a2, n - a2) for m in sorted(check) if ref_counts(n, m, a1, n - a1, a2, n - a2) == want]
            assert got == [s for s in sols if s[2] == a1 and s[4] == a2], (got, sols)
    return sols


def make_kat():
    tuples = []
    for t in KAT:
        tuples.append({"counts": list(t), "expect": ref_counts(*t)})
    for src, n, f1, f2, rsq, dp in PUBLISHED:
        sols = invert_published(n, f1, f2, rsq, dp)
        assert len(sols) == 1, (src, sols)              # the published numbers pin ONE count tuple
        exp = ref_counts(*sols[0])
        assert exp == {"r_square": rsq, "d_prime": dp, "var_1_alt_freq": f1, "var_2_alt_freq": f2}, (src, exp)
        tuples.append({"counts": list(sols[0]), "expect": exp, "published": src})
    for src, n, f1, f2, rsq, dp in NO_SOLUTION:
        assert invert_published(n, f1, f2, rsq, dp) == [], src
    lit = []
    for g1, g2 in LITERAL:
        lit.append({"g1": list(g1), "g2": list(g2), "expect": ref_calc_ld(g1, g2)})
    errors = []
    for g1, g2 in [([], []), ([], [1, 0]), ([1, 0], [])]:
        try:
            ref_calc_ld(g1, g2)
            errors.append({"g1": g1, "g2": g2, "raises": None})
        except Exception as exc:  # noqa: BLE001
            errors.append({"g1": g1, "g2": g2, "raises": type(exc).__name__})
    (OUT / "kat_counts.json").write_text(json.dumps({"tuples": tuples, "literal": lit, "errors": errors}, indent=1))
    print("kat_counts.json:", len(tuples), "tuples,", len(lit), "literal,", len(errors), "errors")


def compositions(total, parts):
    """All tuples of `parts` non-negative ints summing to `total`."""
    if parts == 1:
        yield (total,)
        return
    for first in range(total + 1):
        for rest in compositions(total - first, parts - 1):
            yield (first,) + rest


def make_small_n():
    rows = []
    seen = set()
    for n in (4, 8):
        # all 3x3 joint tables with total n
        for cells in compositions(n, 9):
            c11, c10, c12, c01, c00, c02, c21, c20, _c22 = cells
            t = (n, c11, c11 + c10 + c12, c01 + c00 + c02, c11 + c01 + c21, c10 + c00 + c20)
            if t in seen:
                continue
            seen.add(t)
            rows.append(t)
    for n in (16, 37, 100):
        for a1 in range(n + 1):
            for a2 in range(n + 1):
                for n11 in range(max(0, a1 + a2 - n), min(a1, a2) + 1):
                    rows.append((n, n11, a1, n - a1, a2, n - a2))
    arr = np.array(rows, dtype=np.uint16)
    k_r = np.empty(len(rows), dtype=np.uint32)
    k_d = np.empty(len(rows), dtype=np.uint32)
    k_f1 = np.empty(len(rows), dtype=np.uint16)
    k_f2 = np.empty(len(rows), dtype=np.uint16)
    flags = np.empty(len(rows), dtype=np.uint8)
    for idx, t in enumerate(rows):
        res = ref_counts(*t)
        kr, ir = enc(res["r_square"])
        kd, idp = enc(res["d_prime"])
        k_r[idx], k_d[idx] = kr, kd
        k_f1[idx], k_f2[idx] = enc(res["var_1_alt_freq"])[0], enc(res["var_2_alt_freq"])[0]
        flags[idx] = (orc.FLAG_DPRIME_INT0 if idp else 0) | (orc.FLAG_RSQ_INT0 if ir else 0)
    _reference.save_npz(OUT / "small_n.npz", counts=arr, k_rsq=k_r, k_dp=k_d, k_f1=k_f1, k_f2=k_f2, flags=flags)
    print("small_n.npz:", len(rows), "tuples")


PANELS = {
    # name: (n_snps, n_hap, seed, miss)
    "c1_64x5008": (64, 5008, 7, 0.0),
    "miss_32x5008": (32, 5008, 1, 0.01),
    "eur_64x1008": (64, 1008, 7, 0.0),
    "tie_96x100": (96, 100, 1, 0.0),
    "odd_96x37": (96, 37, 7, 0.02),
}


def panel_lists(codes):
    """int8 codes -> per-variant Python lists as calc_ld receives them; code 2 alternates None / 2."""
    rows = []
    for r in codes:
        lst = []
        for k, v in enumerate(r.tolist()):
            lst.append(v if v != 2 else (None if k % 2 else 2))
        rows.append(lst)
    return rows


def make_panels():
    out = {}
    for name, (n_snps, n_hap, seed, miss) in PANELS.items():
        codes = synth.synth_codes_host(n_snps, n_hap, seed=seed, miss=miss)
        rows = panel_lists(codes)
        npair = n_snps * (n_snps - 1) // 2
        n11 = np.empty(npair, dtype=np.uint16)
        k_r = np.empty(npair, dtype=np.uint32)
        k_d = np.empty(npair, dtype=np.uint32)
        flags = np.empty(npair, dtype=np.uint8)
        k_f = np.empty(n_snps, dtype=np.uint16)
        idx = 0
        for i in range(n_snps):
            for j in range(i):
                res = ref_calc_ld(rows[i], rows[j])          # var_1 = row, var_2 = col (ld_triangle.py:193)
                n11[idx] = orc.pair_counts_lists(rows[i], rows[j])[1]
                kr, ir = enc(res["r_square"])
                kd, idp = enc(res["d_prime"])
                k_r[idx], k_d[idx] = kr, kd
                flags[idx] = (orc.FLAG_DPRIME_INT0 if idp else 0) | (orc.FLAG_RSQ_INT0 if ir else 0)
                if j == 0:
                    k_f[i] = enc(res["var_1_alt_freq"])[0]
                if i == 1:
                    k_f[0] = enc(res["var_2_alt_freq"])[0]
                idx += 1
        out[name + ".n11"] = n11
        out[name + ".k_rsq"] = k_r
        out[name + ".k_dp"] = k_d
        out[name + ".flags"] = flags
        out[name + ".k_freq"] = k_f
        out[name + ".sha256"] = np.frombuffer(hashlib.sha256(codes.tobytes()).digest(), dtype=np.uint8)
        print(name, "pairs", npair, "mean r2 k", k_r.mean())
    _reference.save_npz(OUT / "panels.npz", **out)


def make_drivers():
    n_snps, n_hap, seed, miss = PANELS["c1_64x5008"]
    codes = synth.synth_codes_host(n_snps, n_hap, seed=seed, miss=miss)
    rows = panel_lists(codes)
    tri = {}
    for measure in ("r_square", "d_prime"):
        for thres in (None, 0.2, 0.8):
            tri[f"{measure}|{thres}"] = orc.triangle_lists(rows, measure, thres, calc=ref_calc_ld)
    # irregular ascending positions: mostly 500 bp steps with a few clusters and gaps
    rng = np.random.RandomState(5)
    steps = rng.choice([1, 40, 500, 500, 500, 3000], size=n_snps)
    positions = (1000 + np.cumsum(steps)).tolist()
    area = []
    for (queries, flank, measure, thres) in [
        (list(range(n_snps)), 2000, "r_square", 0.8),
        (list(range(n_snps)), 100000, "r_square", 0.2),
        ([0, 5, 17, 18, 19, 40, 63], 5000, "d_prime", 0.9),
        ([3, 31, 32], 0, "r_square", 0.0),
        ([10, 50], 700, "d_prime", 0.0),
    ]:
        hits = orc.area_lists(rows, positions, queries, flank, measure, thres, calc=ref_calc_ld)
        area.append({"queries": queries, "flank": flank, "measure": measure, "thres": thres,
                     "hits": [list(h) for h in hits]})
        print("area", flank, measure, thres, "hits", len(hits))
    (OUT / "drivers.json").write_text(json.dumps({"panel": "c1_64x5008", "positions": positions, "triangle": tri,
                                                   "area": area}))


if __name__ == "__main__":
    if "--out" in sys.argv:
        OUT = Path(sys.argv[sys.argv.index("--out") + 1])
        OUT.mkdir(parents=True, exist_ok=True)
    make_kat()
    make_small_n()
    make_panels()
    make_drivers()
