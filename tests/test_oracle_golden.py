"""CPU: the oracle (Python and C restatements) against the golden vectors made with the real reference.

This is what pins the oracle: if these fail, nothing the GPU tests say about parity means anything.
"""
import hashlib
import random

import numpy as np
import pytest

from conftest import PANELS, tri_pairs
from oracle import c_oracle
from oracle import ld_oracle as orc


def enc(v):
    if isinstance(v, int) and not isinstance(v, bool):
        return 0, 1
    return round(v * 1e4), 0


def same_typed(a, b):
    """equal AND of the same Python type (int 0 vs float 0.0 matters: it reaches str() in the writers)."""
    return a == b and type(a) is type(b)


# ------------------------------------------------------------------ F1: known answers
def test_py_oracle_kat_tuples(kat):
    for item in kat["tuples"]:
        got = orc.ld_from_counts(*item["counts"])
        for k in orc.KEYS:
            assert same_typed(got[k], item["expect"][k]), (item["counts"], k, got[k], item["expect"][k])


def test_py_oracle_literal_lists(kat):
    for item in kat["literal"]:
        got = orc.calc_ld_lists(item["g1"], item["g2"])
        for k in orc.KEYS:
            assert same_typed(got[k], item["expect"][k]), (item, k, got[k])


def test_py_oracle_errors(kat):
    for item in kat["errors"]:
        assert item["raises"] == "ZeroDivisionError"
        with pytest.raises(ZeroDivisionError):
            orc.calc_ld_lists(item["g1"], item["g2"])


def test_c_oracle_kat_tuples(kat):
    for item in kat["tuples"]:
        rsq_raw, dp_raw, rsq_rnd, dp_rnd, fa1, fa2, flags = c_oracle.ld_from_counts(*item["counts"])
        e = item["expect"]
        assert (round(rsq_rnd * 1e4), 1 if flags & orc.FLAG_RSQ_INT0 else 0) == enc(e["r_square"]), item
        assert (round(dp_rnd * 1e4), 1 if flags & orc.FLAG_DPRIME_INT0 else 0) == enc(e["d_prime"]), item
        assert rsq_rnd == e["r_square"] and dp_rnd == e["d_prime"]
        assert c_oracle.round4(fa1) == e["var_1_alt_freq"] and c_oracle.round4(fa2) == e["var_2_alt_freq"]


def test_published_known_answers_are_in_the_fixture(kat):
    """The reference's only real-data numbers -- README.md:168-193 (EUR, n = 1006: three pairs of chr6) and
    gallery/ld_lite_tabular_output.png -- are F1 rows: the generator inverts each to its unique count tuple with the
    reference itself (tests/golden/make_golden.py, PUBLISHED); here both oracles reproduce the printed values."""
    pub = [t for t in kat["tuples"] if "published" in t]
    readme = {tuple(t["counts"]): t["expect"] for t in pub if t["published"].startswith("README.md:168-193")}
    assert readme == {
        (1006, 439, 590, 416, 742, 264): {"r_square": 0.0003, "d_prime": 0.0247, "var_1_alt_freq": 0.5865, "var_2_alt_freq": 0.7376},
        (1006, 481, 637, 369, 742, 264): {"r_square": 0.0027, "d_prime": 0.0668, "var_1_alt_freq": 0.6332, "var_2_alt_freq": 0.7376},
        (1006, 590, 637, 369, 590, 416): {"r_square": 0.8216, "d_prime": 1.0, "var_1_alt_freq": 0.6332, "var_2_alt_freq": 0.5865}}
    lite = [t for t in pub if t["published"].startswith("gallery/ld_lite_tabular_output.png")]
    assert sorted(t["counts"][0] for t in lite) == [1700, 3160, 3400, 4568, 4860]
    for t in lite:
        assert t["expect"] == {"r_square": 0.7807, "d_prime": 0.9144, "var_1_alt_freq": 0.5247, "var_2_alt_freq": 0.5418}
    for t in pub:
        assert orc.ld_from_counts(*t["counts"]) == t["expect"]
        rnd = c_oracle.ld_from_counts(*t["counts"])
        assert (rnd[2], rnd[3]) == (t["expect"]["r_square"], t["expect"]["d_prime"])


# ------------------------------------------------------------------ F2: exhaustive small n
def test_py_oracle_small_n(small_n):
    counts = small_n["counts"].astype(np.int64)
    # the Python scalar oracle on a deterministic 20k sample plus every n <= 8 tuple
    idx = np.flatnonzero(counts[:, 0] <= 8).tolist()
    rnd = random.Random(3)
    idx += rnd.sample(range(len(counts)), 20000)
    for k in idx:
        got = orc.ld_from_counts(*counts[k].tolist())
        kr, ir = enc(got["r_square"])
        kd, idp = enc(got["d_prime"])
        fl = (orc.FLAG_DPRIME_INT0 if idp else 0) | (orc.FLAG_RSQ_INT0 if ir else 0)
        assert (kr, kd, fl) == (small_n["k_rsq"][k], small_n["k_dp"][k], small_n["flags"][k]), counts[k]
        assert enc(got["var_1_alt_freq"])[0] == small_n["k_f1"][k]
        assert enc(got["var_2_alt_freq"])[0] == small_n["k_f2"][k]


def test_c_oracle_small_n_all(small_n):
    counts = small_n["counts"].astype(np.uint32)
    for n in np.unique(counts[:, 0]):
        m = counts[:, 0] == n
        c = counts[m]
        _, _, rsq_rnd, dp_rnd, flags = c_oracle.ld_from_counts_v(int(n), c[:, 1], c[:, 2], c[:, 3], c[:, 4], c[:, 5])
        assert np.array_equal(np.rint(rsq_rnd * 1e4).astype(np.uint32), small_n["k_rsq"][m])
        assert np.array_equal(np.rint(dp_rnd * 1e4).astype(np.uint32), small_n["k_dp"][m])
        assert np.array_equal(flags, small_n["flags"][m])
        # k / 1e4 is the very double the reference returns
        assert np.array_equal(rsq_rnd, small_n["k_rsq"][m] / 1e4)


def test_c_oracle_dd_vs_pow_within_two_ulp(small_n):
    """pow(d, 2.0) (what CPython's d ** 2 calls) and d*d (what the GPU computes) differ by at most one
    ulp of d^2 (libm pow is not correctly rounded: 103 of these 176 851 tuples), i.e. at most two ulp of
    the quotient r^2 -- 13 orders of magnitude inside the 1e-6 tolerance of the float outputs."""
    counts = small_n["counts"].astype(np.uint32)
    c = counts[counts[:, 0] == 100]
    a = c_oracle.ld_from_counts_v(100, c[:, 1], c[:, 2], c[:, 3], c[:, 4], c[:, 5], libm_pow=True)
    b = c_oracle.ld_from_counts_v(100, c[:, 1], c[:, 2], c[:, 3], c[:, 4], c[:, 5], libm_pow=False)
    assert np.all(np.abs(a[0] - b[0]) <= 2 * np.spacing(np.maximum(a[0], b[0])))
    assert 0 < np.count_nonzero(a[0] != b[0]) < 1000
    assert np.array_equal(a[1], b[1])                 # D' does not involve the square
    assert np.array_equal(a[2], b[2])                 # and the 4-decimal results coincide on this set


# ------------------------------------------------------------------ round4
def test_round4_matches_python_round():
    rnd = random.Random(11)
    vals = [0.0, 0.5, 1.0, 0.03125, 0.84375, 5e-05, 0.00005, 0.21875, 1e-300, 27 / 32, 7 / 32, 2.5e7, 12345.67895]
    vals += [k / 1e4 + 5e-5 for k in range(0, 200)]            # decimal ties as doubles
    vals += [(2 * k + 1) / 32 for k in range(16)]              # exact binary ties
    vals += [rnd.random() for _ in range(20000)]
    vals += [rnd.random() * 10 ** rnd.randint(-8, 6) for _ in range(20000)]
    for v in vals:
        assert orc.round4(v) == round(v, 4), v
        assert c_oracle.round4(v) == round(v, 4), v


# ------------------------------------------------------------------ generator + F3 panels
def test_generator_is_pinned(panel_codes, panels_golden):
    for name, codes in panel_codes.items():
        assert hashlib.sha256(codes.tobytes()).digest() == panels_golden[name + ".sha256"].tobytes(), name


def test_generator_offset_shards():
    from ld_tools_amd import synth

    full = synth.synth_codes_host(200, 77, seed=9, miss=0.01)
    for off, cnt in [(0, 50), (50, 31), (81, 119), (33, 64)]:
        assert np.array_equal(synth.synth_codes_host(cnt, 77, seed=9, miss=0.01, snp_offset=off), full[off:off + cnt])


@pytest.mark.parametrize("name", list(PANELS))
def test_c_oracle_panels(name, panel_codes, panels_golden):
    codes = panel_codes[name]
    p = c_oracle.Panel(codes)
    # packing against an independent numpy packer
    alt_np, ref_np = orc.pack_planes(codes)
    assert np.array_equal(p.alt, alt_np) and np.array_equal(p.ref, ref_np)
    assert np.array_equal(p.acnt, (codes == 1).sum(1)) and np.array_equal(p.rcnt, (codes == 0).sum(1))
    t = p.triangle()
    rows, cols = tri_pairs(p.n_snps)
    assert np.array_equal(t["n11"][rows, cols], panels_golden[name + ".n11"])
    assert np.array_equal(np.rint(t["rsq_rnd"][rows, cols] * 1e4).astype(np.uint32), panels_golden[name + ".k_rsq"])
    assert np.array_equal(np.rint(t["dp_rnd"][rows, cols] * 1e4).astype(np.uint32), panels_golden[name + ".k_dp"])
    assert np.array_equal(t["flags"][rows, cols], panels_golden[name + ".flags"])
    kf = np.array([round(orc.round4(a / p.n_hap) * 1e4) for a in p.acnt], dtype=np.uint16)
    assert np.array_equal(kf, panels_golden[name + ".k_freq"])


def test_numpy_epilogue_matches_c(panel_codes):
    codes = panel_codes["tie_96x100"]
    p = c_oracle.Panel(codes)
    t = p.triangle(libm_pow=False)
    rows, cols = tri_pairs(p.n_snps)
    rsq, dp, fl = orc.ld_raw_from_counts_np(p.n_hap, t["n11"][rows, cols], p.acnt[rows], p.rcnt[rows], p.acnt[cols],
                                            p.rcnt[cols])
    assert np.array_equal(rsq, t["rsq_raw"][rows, cols]) and np.array_equal(dp, t["dp_raw"][rows, cols])
    assert np.array_equal(fl, t["flags"][rows, cols])


# ------------------------------------------------------------------ F4 drivers
def _lists(codes):
    return [[v if v != 2 else None for v in r.tolist()] for r in codes]


def test_py_oracle_triangle_driver(drivers, panel_codes):
    rows = _lists(panel_codes[drivers["panel"]])
    for key, want in drivers["triangle"].items():
        measure, thres = key.split("|")
        thres = None if thres == "None" else float(thres)
        got = orc.triangle_lists(rows, measure, thres)
        assert len(got) == len(want)
        for gr, wr in zip(got, want):
            assert all(same_typed(a, b) for a, b in zip(gr, wr)), key


def test_oracles_area_driver(drivers, panel_codes):
    codes = panel_codes[drivers["panel"]]
    rows = _lists(codes)
    pos = drivers["positions"]
    p = c_oracle.Panel(codes)
    for case in drivers["area"]:
        want = [tuple(h) for h in case["hits"]]
        got = orc.area_lists(rows, pos, case["queries"], case["flank"], case["measure"], case["thres"])
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g[:2] == w[:2] and g[5] == w[5] and all(same_typed(a, b) for a, b in zip(g[2:5], w[2:5])), case
        hq, ho, hr, hd, hf = p.area(pos, case["queries"], case["flank"], 0 if case["measure"] == "r_square" else 1,
                                    case["thres"])
        assert [(int(a), int(b)) for a, b in zip(hq, ho)] == [(w[0], w[1]) for w in want]
        assert np.array_equal(hr, np.array([w[3] for w in want], dtype=np.float64))
        assert np.array_equal(hd, np.array([w[4] for w in want], dtype=np.float64))


# ------------------------------------------------------------------ the pin's own recipe
REFERENCE_CALC_LD = "/root/reference/backend/calc_ld.py"
GOLDEN_FILES = ("kat_counts.json", "small_n.npz", "panels.npz", "drivers.json", "driver_text.json")


@pytest.mark.skipif(not __import__("os").path.isfile(REFERENCE_CALC_LD),
                    reason="the reference exists in the build container only")
def test_golden_generators_load_the_reference_and_reproduce_the_fixtures(tmp_path):
    """Regenerate every fixture from the REAL reference into a temp dir and byte-compare with tests/golden/.

    Guards against (i) the generators resolving ``backend.calc_ld`` to this repo's own top-level ``backend/``
    package (the GPU drop-in) instead of /root/reference/backend/calc_ld.py:3-99 -- which would make the golden
    vectors circular -- and (ii) committed fixtures that are not what the committed scripts produce.
    """
    import subprocess
    import sys
    from pathlib import Path

    golden = Path(__file__).resolve().parent / "golden"
    sys.path.insert(0, str(golden))
    try:
        import _reference
    finally:
        sys.path.remove(str(golden))
    fn = _reference.load_reference_calc_ld()
    assert fn.__code__.co_filename == REFERENCE_CALC_LD
    for script in ("make_golden.py", "make_golden_drivers.py"):
        text = (golden / script).read_text()
        assert "load_reference_calc_ld()" in text and "from backend" not in text, script
        subprocess.run([sys.executable, str(golden / script), "--out", str(tmp_path)], check=True,
                       stdout=subprocess.DEVNULL, timeout=600)
    for name in GOLDEN_FILES:
        assert (tmp_path / name).read_bytes() == (golden / name).read_bytes(), f"{name}: regen differs"


def test_degenerate_snps_give_int_zero_against_any_snp():
    """csrc/ldx_common.h, kSnpDegenerate (round 6): a SNP with no ALT allele, or with no REF allele -- whatever is missing --
    gives the int 0 for BOTH values against ANY other SNP in either argument order -- calc_ld.py:66-69 / 73-76 (the bound is 0
    or -0.0: ZeroDivisionError) and :89-90 -- which is what lets the fp32 tier force such cells without looking at n11.  The
    oracle's list / zip / count path (itself pinned to the reference) on random partners."""
    import random

    from oracle import ld_oracle as orc

    rng = random.Random(6)
    for n in (1, 2, 7, 64, 301):
        degenerate = [[0] * n, [1] * n, [2] * n, [None] * n, [rng.choice((0, 2)) for _ in range(n)],
                      [rng.choice((1, 1, 2, None)) for _ in range(n)], [rng.choice((0, 0, None)) for _ in range(n)]]
        for _ in range(40):
            other = [rng.choice((0, 1, 1, 0, 2, None)) for _ in range(n)]
            for d in degenerate:
                assert d.count(1) == 0 or d.count(0) == 0
                for got in (orc.calc_ld_lists(d, other), orc.calc_ld_lists(other, d)):
                    assert got["r_square"] == 0 and isinstance(got["r_square"], int), (d[:8], other[:8], got)
                    assert got["d_prime"] == 0 and isinstance(got["d_prime"], int), (d[:8], other[:8], got)
