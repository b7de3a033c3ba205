"""GPU: the HIP path (through the C ABI) against the oracle and the golden vectors.

Bar: bit-exact for counts, packed planes, the 4-decimal results (k = value * 10^4) and the int-0 /
float-0.0 flags; 1e-6 (in fact ~1e-15) for the unrounded D' and r^2 floats.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import PANELS, tri_pairs

ROOT = Path(__file__).resolve().parent.parent

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    import ld_tools_amd  # noqa: F401  (raises if libldx.so is missing: no fallback)
    from ld_tools_amd import _lib

    buf = __import__("ctypes").create_string_buffer(64)
    _lib.check(_lib.lib.ldx_device_arch(0, buf, 64))
    assert buf.value.decode().startswith("gfx950"), buf.value
    return torch.device("cuda", 0)


@pytest.fixture(params=["popcount", "mfma", "fp4"])
def path(request, gpu):
    """Run a triangle test once per kernel: AND+BCNT on the VALU, int8 MFMA, FP4 MFMA (the default)."""
    from ld_tools_amd import ops

    ops.set_triangle_path(request.param)
    yield request.param
    ops.set_triangle_path("auto")


def untile(plane_u8: np.ndarray, n_snps: int, n_hap: int) -> np.ndarray:
    """tiled plane bytes -> bool [n_snps][n_hap]"""
    slabs, chunks = (n_snps + 127) // 128, (n_hap + 255) // 256 * 2   # chunks are allocated in pairs
    by = plane_u8.reshape(slabs, chunks, 128, 16).transpose(0, 2, 1, 3).reshape(slabs * 128, chunks * 16)
    bits = np.unpackbits(by, axis=1, bitorder="little")
    return bits[:n_snps, :n_hap].astype(bool), bits


@pytest.fixture(params=["popcount", "mfma", "fp4"])
def area_path(request, gpu):
    """Run an ld_area test once per kernel (the popcount scan of query rows and the matrix-pipe band)."""
    from ld_tools_amd import ops
    ops.set_area_path(request.param)
    yield request.param
    ops.set_area_path("auto")


def k_of(x32: np.ndarray) -> np.ndarray:
    return np.rint(x32.astype(np.float64) * 1e4).astype(np.int64)


def int0_of(x32: np.ndarray) -> np.ndarray:
    return np.signbit(x32)


def flags_of(ld32: np.ndarray) -> np.ndarray:
    return (int0_of(ld32[:, 1]).astype(np.uint8) * 1) | (int0_of(ld32[:, 0]).astype(np.uint8) * 2)


def check_cells(k_true: np.ndarray, flags_true: np.ndarray, ld32: np.ndarray, k16: np.ndarray, tag=None):
    """Both cell formats against the exact k [m, 2] (r_square, d_prime) and the int-0 flags: every value the format
    can hold is exact, every other one is the format's escape -- never a wrong number."""
    k_true = np.asarray(k_true, dtype=np.float64).reshape(-1, 2)
    int0 = np.stack([(flags_true & 2) != 0, (flags_true & 1) != 0], axis=1)
    ld32 = np.asarray(ld32, dtype=np.float32).reshape(-1, 2)
    big32 = (k_true >= 1.024e7) & ~int0
    assert np.array_equal(np.isnan(ld32), big32), tag
    assert np.array_equal(ld32.view(np.uint32)[big32], np.full(int(big32.sum()), 0x7FC00B16, np.uint32)), tag
    assert np.array_equal(np.signbit(ld32) & ~big32, int0), tag
    ok = ~big32
    assert np.array_equal(np.rint(ld32[ok].astype(np.float64) * 1e4), np.where(int0, 0, k_true)[ok]), tag
    assert np.array_equal(ld32[ok], np.where(int0, -0.0, k_true / 1e4).astype(np.float32)[ok]), tag   # the nearest float32
    u = np.asarray(k16).reshape(-1, 2).view(np.uint16).astype(np.int64)
    big16 = (k_true >= 32767) & ~int0
    assert np.array_equal(u == 0x7FFF, big16), tag
    assert np.array_equal(u == 0x8000, int0), tag
    rest = ~big16 & ~int0
    assert np.array_equal(u[rest], k_true[rest].astype(np.int64)), tag


def oracle_rows_against_cells(o, res, ld32, n11, bands, workers=8):
    """Rows [r0, r1) of every band in `bands` -- ALL their cells (row > col) -- against the C oracle (oracle/ld_oracle.c:
    AND + popcount, then calc_ld.py:33-97 op for op): n11 bit-exact, k of r^2 and D' exact, int-0 flags exact.  The oracle
    does ~1.5e7 pairs/s per thread and releases the GIL (ctypes), so the bands are spread over a small thread pool.
    ld32 / n11: host copies of the result's cells (numpy).  Returns the number of cells compared."""
    from concurrent.futures import ThreadPoolExecutor

    def one(band):
        r0, r1 = band
        t = o.triangle_band(r0, r1, libm_pow=True, want=("n11", "rsq_rnd", "dp_rnd", "flags"))   # [r1 - r0][n]
        counts = np.arange(r0, r1, dtype=np.int64)                       # row i has i cells
        rows = np.repeat(counts, counts)
        cols = np.concatenate([np.arange(i, dtype=np.int64) for i in range(r0, r1)])
        idx = res.cell_index(rows, cols)
        rb = rows - r0
        assert np.array_equal(n11[idx], t["n11"][rb, cols]), ("n11", band)
        cells = ld32[idx]
        assert np.array_equal(k_of(cells[:, 0]), np.rint(t["rsq_rnd"][rb, cols] * 1e4).astype(np.int64)), ("r_square", band)
        assert np.array_equal(k_of(cells[:, 1]), np.rint(t["dp_rnd"][rb, cols] * 1e4).astype(np.int64)), ("d_prime", band)
        assert np.array_equal(flags_of(cells), t["flags"][rb, cols]), ("int-0 flags", band)
        return len(rows)

    bands = [b for b in bands if b[1] > max(b[0], 1)]
    with ThreadPoolExecutor(max_workers=workers) as pool:
        return sum(pool.map(one, bands))



# ------------------------------------------------------------------ packing
@pytest.mark.parametrize("name", list(PANELS))
def test_pack_matches_oracle(gpu, name, panel_codes):
    from ld_tools_amd import PackedPanel
    from oracle import c_oracle

    codes = panel_codes[name]
    p = PackedPanel.from_codes(codes)
    o = c_oracle.Panel(codes)
    alt, alt_all = untile(p.alt.cpu().numpy(), p.n_snps, p.n_hap)
    ref, ref_all = untile(p.ref.cpu().numpy(), p.n_snps, p.n_hap)
    assert np.array_equal(alt, codes == 1) and np.array_equal(ref, codes == 0)
    assert alt_all.sum() == (codes == 1).sum() and ref_all.sum() == (codes == 0).sum()   # pad bits / rows are zero
    assert np.array_equal(p.alt_counts(), o.acnt) and np.array_equal(p.ref_counts(), o.rcnt)
    npad = p.padded_snps
    assert not p.acnt[p.n_snps:npad].any() and not p.fa[p.n_snps:npad].any()
    n = float(p.n_hap)
    assert np.array_equal(p.fa[: p.n_snps].cpu().numpy(), o.acnt / n)       # one IEEE division each
    assert np.array_equal(p.fr[: p.n_snps].cpu().numpy(), o.rcnt / n)
    assert np.array_equal(p.q[: p.n_snps].cpu().numpy(), (o.acnt / n) * (o.rcnt / n))
    want_f4 = np.array([round(a / n, 4) for a in o.acnt.tolist()])
    assert np.array_equal(p.alt_freq4().cpu().numpy(), want_f4)


@pytest.mark.parametrize("shape", [(130, 300), (257, 5008), (5, 31), (128, 256), (129, 257), (300, 10240)])
def test_pack_every_byte_value_and_repack(gpu, shape):
    """list.count semantics for ANY int8 code (calc_ld.py:37-40: only == 1 and == 0 count), on shapes either side of the
    kernel's 128-row x 256-haplotype tiles; packing twice into the same panel gives the same planes and counts (the
    counts are accumulated with atomics into vectors the call zeroes itself)."""
    import torch

    from ld_tools_amd import PackedPanel

    n, h = shape
    rng = np.random.RandomState(n * 7 + h)
    codes = rng.randint(-128, 128, size=(n, h)).astype(np.int8)
    codes[rng.random_sample((n, h)) < 0.6] = 1
    codes[rng.random_sample((n, h)) < 0.3] = 0
    p = PackedPanel.from_codes(codes)
    alt, alt_all = untile(p.alt.cpu().numpy(), n, h)
    ref, ref_all = untile(p.ref.cpu().numpy(), n, h)
    assert np.array_equal(alt, codes == 1) and np.array_equal(ref, codes == 0)
    assert alt_all.sum() == (codes == 1).sum() and ref_all.sum() == (codes == 0).sum()
    assert np.array_equal(p.alt_counts(), (codes == 1).sum(axis=1)) and np.array_equal(p.ref_counts(), (codes == 0).sum(axis=1))
    before = (p.alt.clone(), p.ref.clone(), p.acnt.clone(), p.rcnt.clone())
    p.alt.fill_(255)                          # stale contents must not survive a re-pack
    p.pack_from(torch.from_numpy(codes).to(gpu))
    assert all(torch.equal(a, b) for a, b in zip(before, (p.alt, p.ref, p.acnt, p.rcnt)))


def test_pack_unaligned_rows_and_tile_plane(gpu):
    import torch

    from ld_tools_amd import PackedPanel, _lib
    from ld_tools_amd.panel import _stream_ptr
    from oracle import ld_oracle as orc

    rng = np.random.RandomState(2)
    codes = rng.choice(np.array([0, 1, 2], dtype=np.int8), size=(131, 333), p=[0.5, 0.45, 0.05])
    p = PackedPanel.from_codes(codes)          # rows of 333 bytes: not 16-byte aligned -> byte path
    alt, _ = untile(p.alt.cpu().numpy(), 131, 333)
    assert np.array_equal(alt, codes == 1)
    # row-major bit plane -> tiled (ldx_tile_plane_dev)
    alt64, _ = orc.pack_planes(codes)
    rm = torch.from_numpy(alt64.view(np.uint32).astype(np.uint32).view(np.int32).copy()).to(gpu)
    tiled = torch.empty_like(p.alt)
    cnt = torch.empty_like(p.acnt)
    _lib.check(_lib.lib.ldx_tile_plane_dev(rm.data_ptr(), 131, 333, rm.shape[1], tiled.data_ptr(), cnt.data_ptr(),
                                           _stream_ptr()))
    assert torch.equal(tiled, p.alt) and torch.equal(cnt, p.acnt)


def test_synth_device_equals_host(gpu):
    from ld_tools_amd import synth

    for (n, h, seed, miss, off) in [(96, 5008, 20261003, 0.0, 0), (70, 1008, 7, 0.01, 0), (50, 333, 1, 0.02, 45)]:
        dev = synth.synth_codes_device(n, h, seed=seed, miss=miss, snp_offset=off).cpu().numpy()
        host = synth.synth_codes_host(n, h, seed=seed, miss=miss, snp_offset=off)
        assert np.array_equal(dev, host), (n, h, seed)
    # round 6: panels with monomorphic rows and with missing codes confined to a share of the rows (bench.py's odd-panel legs)
    for (n, h, seed, miss, off, mono, mrows) in [(200, 1008, 20261003, 0.0, 0, 0.3, 1.0), (160, 517, 5, 0.001, 37, 0.0, 0.2),
                                                (130, 300, 9, 0.05, 64, 0.25, 0.5)]:
        dev = synth.synth_codes_device(n, h, seed=seed, miss=miss, snp_offset=off, mono=mono, miss_rows=mrows).cpu().numpy()
        host = synth.synth_codes_host(n, h, seed=seed, miss=miss, snp_offset=off, mono=mono, miss_rows=mrows)
        assert np.array_equal(dev, host), (n, h, seed, mono, mrows)
        flat = (dev == dev[:, :1]).all(axis=1)
        if mono:
            assert 0.1 * n < flat.sum() < 0.5 * n, flat.sum()
        if miss and mrows < 1.0:
            with_miss = (dev == 2).any(axis=1)
            assert 0 < with_miss.sum() < 0.8 * n


# ------------------------------------------------------------------ counts (bit-exact contract)
@pytest.mark.parametrize("name", list(PANELS))
def test_pair_counts_match_golden(gpu, name, panel_codes, panels_golden):
    from ld_tools_amd import PackedPanel, pair_counts

    p = PackedPanel.from_codes(panel_codes[name])
    n11 = pair_counts(p).cpu().numpy().view(np.uint32)
    rows, cols = tri_pairs(p.n_snps)
    assert np.array_equal(n11[rows, cols], panels_golden[name + ".n11"])     # vs the reference itself
    assert np.array_equal(n11, n11.T)
    assert np.array_equal(np.diag(n11), p.alt_counts())


def test_pair_counts_rectangular_two_panels(gpu):
    from ld_tools_amd import PackedPanel, pair_counts, synth
    from oracle import c_oracle

    a = synth.synth_codes_host(300, 1008, seed=5, miss=0.01)
    b = synth.synth_codes_host(137, 1008, seed=6, miss=0.0)
    got = pair_counts(PackedPanel.from_codes(a), PackedPanel.from_codes(b)).cpu().numpy().view(np.uint32)
    o = c_oracle.Panel(np.concatenate([a, b]))
    assert np.array_equal(got, o.pair_counts(0, 300, 300, 437))


# ------------------------------------------------------------------ epilogue (vs the reference's own outputs)
def test_epilogue_small_n_exhaustive(gpu, small_n):
    from ld_tools_amd import ld_from_counts
    from oracle import c_oracle

    counts = small_n["counts"].astype(np.uint32)
    for n in np.unique(counts[:, 0]):
        m = counts[:, 0] == n
        c = counts[m]
        raw, rnd, flags, k, k16, sure32 = ld_from_counts(int(n), c[:, 1], c[:, 2], c[:, 3], c[:, 4], c[:, 5], full=True)
        rnd = rnd.cpu().numpy()
        k = k.cpu().numpy()
        assert np.array_equal(k[:, 0], small_n["k_rsq"][m]) and np.array_equal(k[:, 1], small_n["k_dp"][m])   # exact k
        check_cells(k, small_n["flags"][m], rnd, k16.cpu().numpy(), int(n))  # both formats; a tier mismatch poisons
        small = (small_n["k_rsq"][m] < 1e7) & (small_n["k_dp"][m] < 1e7)
        assert small.all()
        assert np.array_equal(k_of(rnd[:, 0]), small_n["k_rsq"][m])          # 4-decimal r^2, exact
        assert np.array_equal(k_of(rnd[:, 1]), small_n["k_dp"][m])           # 4-decimal D', exact
        assert np.array_equal(flags.cpu().numpy(), small_n["flags"][m])      # int 0 vs float 0.0
        assert np.array_equal(flags_of(rnd), small_n["flags"][m])            # ... also encoded as -0.0f
        assert np.array_equal(rnd[:, 0], (small_n["k_rsq"][m] / 1e4).astype(np.float32))
        o = c_oracle.ld_from_counts_v(int(n), c[:, 1], c[:, 2], c[:, 3], c[:, 4], c[:, 5], libm_pow=True)
        raw = raw.cpu().numpy()
        assert np.array_equal(raw[:, 1], o[1])                               # D' is bit-identical
        assert np.allclose(raw[:, 0], o[0], rtol=1e-15, atol=0)              # r^2: d*d vs pow(d,2): <= 2 ulp
        assert np.max(np.abs(raw[:, 0] - o[0]), initial=0) <= 1e-6


@pytest.mark.parametrize("n", [5008, 1008, 2504])
def test_epilogue_random_tuples_at_panel_sizes(gpu, n):
    """Two million count tuples at the real haplotype counts against the C restatement (itself pinned to the
    reference's outputs): common and rare alleles, singletons, monomorphic variants, missing codes (a + r < n), pairs
    at |Dn| <= 2 where the reference's cancellation error is largest, and exact-tie candidates.  ld_from_counts
    runs every production epilogue variant on each tuple and poisons the result if they disagree."""
    from ld_tools_amd import ld_from_counts
    from oracle import c_oracle

    rng = np.random.RandomState(n)
    m = 2_000_000
    kind = rng.randint(0, 6, m)
    u = rng.rand(m)
    a1 = np.where(kind == 1, rng.randint(0, 4, m), (np.sin(np.pi / 2 * u) ** 2 * n).astype(np.int64))
    a2 = np.where(kind == 2, rng.randint(0, 4, m), (np.sin(np.pi / 2 * rng.rand(m)) ** 2 * n).astype(np.int64))
    miss1 = np.where(kind == 3, rng.randint(0, n // 3, m), 0)
    miss2 = np.where(kind == 3, rng.randint(0, n // 3, m), 0)
    a1 = np.minimum(a1, n - miss1)
    a2 = np.minimum(a2, n - miss2)
    r1, r2 = n - miss1 - a1, n - miss2 - a2
    lo = np.maximum(0, a1 + a2 - n)                      # feasible n11 range (ignoring missing: widened below)
    hi = np.minimum(a1, a2)
    indep = np.rint(a1.astype(np.float64) * a2 / n).astype(np.int64)
    n11 = np.where(kind == 4, indep + rng.randint(-2, 3, m), lo + (rng.rand(m) * (hi - lo + 1)).astype(np.int64))
    n11 = np.where(kind == 5, rng.choice([0, 1], m) * hi + (1 - rng.choice([0, 1], m)) * lo, n11)
    n11 = np.clip(n11, np.where(kind == 3, 0, lo), hi)
    arrs = [x.astype(np.uint32) for x in (n11, a1, r1, a2, r2)]
    raw, rnd, flags, k, k16, sure32 = ld_from_counts(n, *arrs, full=True)
    rnd, flags, k = rnd.cpu().numpy(), flags.cpu().numpy(), k.cpu().numpy()
    # the fp32 tier keeps most ordinary pairs (the rest go to the fp64 tier); whenever it keeps one its cell equals the
    # others' (checked in the kernel: a disagreement poisons the cell)
    # (ordinary since round 6, csrc/ldx_common.h snp_class: polymorphic with at most r / 8 missing codes)
    a1_, r1_, a2_, r2_ = (x.astype(np.int64) for x in arrs[1:])
    ordinary = (a1_ > 0) & (r1_ > 0) & (a2_ > 0) & (r2_ > 0) & (8 * (n - a1_ - r1_) <= r1_) & (8 * (n - a2_ - r2_) <= r2_)
    sure32 = sure32.cpu().numpy()
    assert not sure32[~ordinary].any() and sure32[ordinary].mean() > 0.9, sure32[ordinary].mean()
    o_rsq_raw, o_dp_raw, o_rsq, o_dp, o_flags = c_oracle.ld_from_counts_v(n, *arrs, libm_pow=True)
    assert np.array_equal(flags, o_flags)
    # k itself (a double: exact for any magnitude) against the oracle's round(x, 4); k / 1e4 is that very double
    assert np.array_equal(k[:, 0] / 1e4, o_rsq) and np.array_equal(k[:, 1] / 1e4, o_dp)
    # both cell formats: exact where they can hold the value, their escape where they cannot; a disagreement
    # between the epilogue tiers would poison the cell (plain NaN / 0xFFFF) and fail here
    check_cells(k, o_flags, rnd, k16.cpu().numpy(), n)
    small = (o_rsq < 1000.0) & (o_dp < 1000.0)
    assert small.mean() > 0.99 and (~small).any()
    raw = raw.cpu().numpy()
    assert np.array_equal(raw[:, 1], o_dp_raw)                                # unrounded D': bit-identical
    assert np.max(np.abs(raw[:, 0] - o_rsq_raw) / np.maximum(1.0, np.abs(o_rsq_raw))) <= 1e-6


@pytest.mark.parametrize("n", [5008, 1008, 10240, 4096])
def test_epilogue_tuples_with_a_few_missing_codes(gpu, n):
    """Round 6: the fast tiers take polymorphic SNPs with a FEW missing codes (m = n - a - r <= r / 8; csrc/ldx_common.h,
    snp_class) on their common path -- D' and r^2 may then exceed 1 (< 1.27), which the bounds allow for.  Two million such
    tuples -- every feasible kind of n11: the whole range [max(0, a1 + a2 - n), min(a1, a2)], its two ends, and |Dn| small --
    through every tier (ld_from_counts poisons a cell on which two tiers disagree) against the C restatement of
    calc_ld.py:33-97; the fp32 tier must keep most of them."""
    from ld_tools_amd import ld_from_counts
    from oracle import c_oracle

    rng = np.random.RandomState(n + 1)
    m = 2_000_000
    kind = rng.randint(0, 4, m)

    def snp():
        miss = (rng.rand(m) ** 3 * (n // 9)).astype(np.int64)                  # mostly a handful, up to ~n / 9
        a = 1 + (np.sin(np.pi / 2 * rng.rand(m)) ** 2 * (n - miss - 2)).astype(np.int64)
        r = n - miss - a
        ok = (r > 0) & (8 * miss <= r)
        miss = np.where(ok, miss, 0)
        a = np.where(ok, a, np.clip(a, 1, n - 1))
        return a, n - miss - a

    a1, r1 = snp()
    a2, r2 = snp()
    assert ((r1 > 0) & (r2 > 0) & (8 * (n - a1 - r1) <= r1) & (8 * (n - a2 - r2) <= r2)).all()
    lo, hi = np.maximum(0, a1 + a2 - n), np.minimum(a1, a2)
    indep = np.rint(a1.astype(np.float64) * a2 / n).astype(np.int64)
    n11 = np.where(kind == 0, lo + (rng.rand(m) * (hi - lo + 1)).astype(np.int64),
                   np.where(kind == 1, indep + rng.randint(-2, 3, m), np.where(kind == 2, lo, hi)))
    n11 = np.clip(n11, lo, hi)
    arrs = [x.astype(np.uint32) for x in (n11, a1, r1, a2, r2)]
    raw, rnd, flags, k, k16, sure32 = ld_from_counts(n, *arrs, full=True)
    rnd, flags, k, sure32 = rnd.cpu().numpy(), flags.cpu().numpy(), k.cpu().numpy(), sure32.cpu().numpy()
    assert sure32.mean() > 0.9, sure32.mean()
    o_rsq_raw, o_dp_raw, o_rsq, o_dp, o_flags = c_oracle.ld_from_counts_v(n, *arrs, libm_pow=True)
    assert np.array_equal(flags, o_flags)
    assert np.array_equal(k[:, 0] / 1e4, o_rsq) and np.array_equal(k[:, 1] / 1e4, o_dp)
    check_cells(k, o_flags, rnd, k16.cpu().numpy(), n)
    assert max(o_rsq.max(), o_dp.max()) < 1.27 and max(o_rsq.max(), o_dp.max()) > 1.0      # beyond 1, inside the bound
    assert np.array_equal(raw.cpu().numpy()[:, 1], o_dp_raw)


def test_epilogue_kat(gpu, kat):
    from ld_tools_amd import ld_from_counts

    for item in kat["tuples"]:
        n, n11, a1, r1, a2, r2 = item["counts"]
        raw, rnd, flags, k, k16, _ = ld_from_counts(n, [n11], [a1], [r1], [a2], [r2], full=True)
        e = item["expect"]
        f = int(flags[0])
        # k / 10^4 in double IS the reference's round(x, 4), whatever the magnitude (the D' >> 1 tuples included)
        got_r = 0 if f & 2 else float(k[0, 0]) / 1e4
        got_d = 0 if f & 1 else float(k[0, 1]) / 1e4
        assert got_r == e["r_square"] and type(got_r) is type(e["r_square"]), item
        assert got_d == e["d_prime"] and type(got_d) is type(e["d_prime"]), item
        check_cells(k.cpu().numpy(), flags.cpu().numpy(), rnd.cpu().numpy(), k16.cpu().numpy(), item)


# ------------------------------------------------------------------ ld_triangle
@pytest.mark.parametrize("name", list(PANELS))
def test_triangle_matches_golden(gpu, path, name, panel_codes, panels_golden):
    from ld_tools_amd import PackedPanel, ld_triangle
    from oracle import c_oracle

    codes = panel_codes[name]
    p = PackedPanel.from_codes(codes)
    res = ld_triangle(p, want_raw=True, want_n11=True)
    rows, cols = tri_pairs(p.n_snps)
    idx = res.cell_index(rows, cols)
    ld32 = res.ld32.cpu().numpy()[idx]
    assert np.array_equal(res.n11.cpu().numpy().view(np.uint32)[idx], panels_golden[name + ".n11"])
    assert np.array_equal(k_of(ld32[:, 0]), panels_golden[name + ".k_rsq"])
    assert np.array_equal(k_of(ld32[:, 1]), panels_golden[name + ".k_dp"])
    assert np.array_equal(flags_of(ld32), panels_golden[name + ".flags"])
    o = c_oracle.Panel(codes).triangle(libm_pow=True)
    raw = res.raw.cpu().numpy()[idx]
    assert np.max(np.abs(raw[:, 0] - o["rsq_raw"][rows, cols])) <= 1e-6
    assert np.array_equal(raw[:, 1], o["dp_raw"][rows, cols])
    # every other cell of the strip output (row <= col, pad rows) is zero
    mask = np.ones(len(res.ld32), dtype=bool)
    mask[idx] = False
    assert not res.ld32.cpu().numpy()[mask].any()


def test_triangle_dense_and_thresholds(gpu, path, drivers, panel_codes):
    from ld_tools_amd import PackedPanel, ld_triangle

    p = PackedPanel.from_codes(panel_codes[drivers["panel"]])
    res = ld_triangle(p)
    for key, want in drivers["triangle"].items():
        measure, thres = key.split("|")
        thres = None if thres == "None" else float(thres)
        dense = res.dense(measure, thres).cpu().numpy()
        for i, wrow in enumerate(want):
            for j, w in enumerate(wrow):
                g = dense[i, j]
                if isinstance(w, int):
                    assert g == 0 and np.signbit(g), (key, i, j)          # int 0 <-> -0.0f
                else:
                    assert not (g == 0 and np.signbit(g)) and round(float(g), 4) == w, (key, i, j, g, w)


def _panel_with_huge_values(n_extra=70, h=5008):
    """A panel whose pair (1, 0) is the known-answer tuple (5008, 40, 40, 4000, 60, 1) of the reference:
    r_square 4080.4507, d_prime 4948.0 (a + r < n with a vanishing bound), among ordinary SNPs and a second
    missing-code row that gives values between 3.2767 and 1024."""
    from conftest import realise
    from ld_tools_amd import synth

    g1, g2 = realise(5008, 40, 40, 4000, 60, 1)
    codes = synth.synth_codes_host(n_extra + 3, h, seed=33, miss=0.0)
    codes[0] = np.array(g2, dtype=np.int8)         # var_2 of the tuple (the column)
    codes[1] = np.array(g1, dtype=np.int8)         # var_1 (the row)
    g3, g4 = realise(5008, 30, 60, 4000, 40, 400)  # a milder one on rows (3, 2)
    codes[2] = np.array(g4, dtype=np.int8)
    codes[3] = np.array(g3, dtype=np.int8)
    return codes


def test_triangle_k16_cells_equal_ld32_cells(gpu, path):
    """The 4-byte cell format carries the same (k, int-0) as the 8-byte one for every cell either can hold, on all
    three kernels, with missing codes, monomorphic rows and values beyond both formats in the panel."""
    import torch
    from ld_tools_amd import PackedPanel, ld_triangle, ops
    from oracle import c_oracle

    codes = _panel_with_huge_values()
    codes[10] = 0
    codes[11, ::5] = 2
    p = PackedPanel.from_codes(codes)
    a = ld_triangle(p, want_n11=True)
    b = ld_triangle(p, want_n11=True, fmt="k16")
    assert b.k16.dtype == torch.int16 and b.ld32 is None and b.fmt == "k16"
    n = p.n_snps
    rows, cols = np.tril_indices(n, -1)
    idx, idx16 = a.cell_index(rows, cols), b.cell_index(rows, cols)   # each format has its own order inside a unit (include/ldx.h)
    assert np.array_equal(a.n11.cpu().numpy()[idx], b.n11.cpu().numpy()[idx16])
    o = c_oracle.Panel(codes).triangle(libm_pow=True, want=("rsq_rnd", "dp_rnd", "flags"))
    k_true = np.stack([np.rint(o["rsq_rnd"][rows, cols] * 1e4), np.rint(o["dp_rnd"][rows, cols] * 1e4)], axis=1)
    check_cells(k_true, o["flags"][rows, cols], a.ld32.cpu().numpy()[idx], b.k16.cpu().numpy()[idx16], path)
    assert (k_true >= 1.024e7).any() and ((k_true >= 32767) & (k_true < 1.024e7)).any()
    # cells outside the triangle are zero in both formats, and so are their counts
    mask, mask16 = np.ones(len(b.k16), dtype=bool), np.ones(len(b.k16), dtype=bool)
    mask[idx] = False
    mask16[idx16] = False
    assert not b.k16.cpu().numpy()[mask16].any() and not a.ld32.cpu().numpy()[mask].any()
    assert not b.n11.cpu().numpy()[mask16].any() and not a.n11.cpu().numpy()[mask].any()
    # k_and_int0 decodes both the same way
    ka, za, ea = a.k_and_int0(idx)
    kb, zb, eb = b.k_and_int0(idx16)
    both = ~ea & ~eb
    assert np.array_equal(ka[both], kb[both]) and np.array_equal(za, zb) and (ea <= eb).all()
    # the dense matrices agree wherever neither is an escape; k16's escapes are a superset
    for measure, thres in (("r_square", None), ("d_prime", 0.3), ("r_square", 0.05)):
        da, db = a.dense(measure, thres).cpu().numpy(), b.dense(measure, thres).cpu().numpy()
        ok = ~np.isnan(da) & ~np.isnan(db)
        assert np.array_equal(da[ok].view(np.uint32), db[ok].view(np.uint32)), (measure, thres)
        assert (np.isnan(da) <= np.isnan(db)).all() and np.isnan(da).any()
        # ... and the escapes resolve to the reference's values in both
        va, fa_ = a.dense_values(measure, thres)
        vb, fb_ = b.dense_values(measure, thres)
        col = o["rsq_rnd"] if measure == "r_square" else o["dp_rnd"]
        for fixes in (fa_, fb_):
            assert fixes
            for (i, j), val in fixes.items():
                want = float(col[i, j]) if (thres is None or col[i, j] >= thres) else 0
                assert val == want and type(val) is type(want), (measure, thres, i, j, val, want)
        assert {k: v for k, v in fb_.items() if k in fa_} == fa_
    assert a.dense_values("r_square")[1][(1, 0)] == 4080.4507 and a.dense_values("d_prime")[1][(1, 0)] == 4948.0


@pytest.mark.parametrize("n,h,kw", [(10000, 5008, {}), (50000, 1008, {}), (3000, 300, dict(miss=0.01)),
                                    (6000, 1008, dict(mono=0.3, miss=0.001, miss_rows=0.2)), (700, 5008, {})])
def test_one_measure_cells_equal_the_two_value_cells(gpu, n, h, kw):
    """the one-measure formats (2 bytes per pair: what a table writer needs, ld_triangle.py:223-230,
    344-360 print ONE measure) skip the other value's arithmetic in the fp32 tier.  For EVERY pair of configs[1] and
    configs[4] (and panels with missing codes, monomorphic SNPs, all-short passes) the r_square cell equals the r_square half
    of the two-value kernel's 4-byte cell and the d_prime cell its d_prime half -- FP4 kernel and popcount kernel -- escapes
    and int-0 marks included."""
    import torch
    from ld_tools_amd import PackedPanel, ld_triangle, synth

    p = PackedPanel.from_codes(synth.synth_codes_device(n, h, seed=synth.BENCH_SEED, **kw))
    both = ld_triangle(p, fmt="k16", path="fp4")
    for col, fmt in ((0, "k16r"), (1, "k16d")):
        one = ld_triangle(p, fmt=fmt, path="fp4")
        assert one.k16one.dtype == torch.int16 and one.k16one.numel() == both.k16.shape[0]
        assert torch.equal(one.k16one, both.k16[:, col].contiguous()), (fmt, "fp4")
        if n <= 10000:
            pop = ld_triangle(p, fmt=fmt, path="popcount")
            assert torch.equal(pop.k16one, one.k16one), (fmt, "popcount")
            del pop
        del one
    # the int8 matrix kernel does not carry these formats and says so; side outputs are refused
    from ld_tools_amd import _lib
    with pytest.raises(_lib.LdxError):
        ld_triangle(p, fmt="k16r", path="mfma")
    with pytest.raises(_lib.LdxError):
        ld_triangle(p, fmt="k16d", want_n11=True)
    if n <= 3000:   # dense() and its escape resolution (values >= 3.2767 only arise with missing codes)
        for measure, fmt in (("r_square", "k16r"), ("d_prime", "k16d")):
            one = ld_triangle(p, fmt=fmt)
            d1, f1 = one.dense_values(measure, 0.2)
            d2, f2 = both.dense_values(measure, 0.2)
            assert np.array_equal(d1.view(np.int32), d2.view(np.int32)) and f1 == f2
            with pytest.raises(_lib.LdxError):
                one.dense("d_prime" if measure == "r_square" else "r_square")


def test_ld_pairs_exact_for_any_magnitude(gpu):
    from ld_tools_amd import PackedPanel, ops
    from oracle import c_oracle

    codes = _panel_with_huge_values()
    p = PackedPanel.from_codes(codes)
    rng = np.random.RandomState(5)
    rows = np.concatenate([[1, 3, 0, 2], rng.randint(0, p.n_snps, 500)])
    cols = np.concatenate([[0, 2, 1, 3], rng.randint(0, p.n_snps, 500)])      # any order, repeats, row == col too
    got = ops.ld_pairs(p, rows, cols)
    oc = c_oracle.Panel(codes)
    n11 = np.array([oc.pair_counts(int(r), int(r) + 1, int(c), int(c) + 1)[0, 0] for r, c in zip(rows, cols)], np.uint32)
    assert np.array_equal(got["n11"], n11)
    rsq_raw, dp_raw, rsq, dp, flags = c_oracle.ld_from_counts_v(p.n_hap, n11, oc.acnt[rows], oc.rcnt[rows], oc.acnt[cols],
                                                                oc.rcnt[cols], libm_pow=True)
    assert np.array_equal(got["k"][:, 0] / 1e4, rsq) and np.array_equal(got["k"][:, 1] / 1e4, dp)
    assert np.array_equal(got["flags"], flags)
    assert np.array_equal(got["raw"][:, 1], dp_raw)
    assert got["k"][0, 0] == 40804507 and got["k"][0, 1] == 49480000
    with pytest.raises(Exception):
        ops.ld_pairs(p, [0, 1], [0])


def test_triangle_sharded_units_equal_full(gpu, path):
    from ld_tools_amd import PackedPanel, dist, ld_triangle, synth

    p = PackedPanel.from_codes(synth.synth_codes_device(700, 1008, seed=4, miss=0.005))
    full = ld_triangle(p, want_n11=True)
    for world in (2, 3, 8):
        parts = dist.unit_partition(700, world)
        pieces = [ld_triangle(p, unit_range=r, want_n11=True) for r in parts]
        import torch
        assert torch.equal(torch.cat([x.ld32 for x in pieces]).view(torch.int32), full.ld32.view(torch.int32))
        assert torch.equal(torch.cat([x.n11 for x in pieces]), full.n11)


@pytest.mark.parametrize("n_snps,n_hap", [(1, 64), (2, 1), (2, 128), (9, 129), (129, 37), (257, 2000), (130, 10240)])
def test_triangle_edge_shapes(gpu, path, n_snps, n_hap):
    from ld_tools_amd import PackedPanel, ld_triangle, synth
    from oracle import c_oracle

    codes = synth.synth_codes_host(n_snps, n_hap, seed=12, miss=0.03)
    codes[0] = 1                      # a monomorphic ALT variant
    if n_snps > 2:
        codes[2] = 0                  # and a monomorphic REF one
    p = PackedPanel.from_codes(codes)
    res = ld_triangle(p, want_raw=True, want_n11=True)
    rows, cols = tri_pairs(n_snps)
    if n_snps == 1:
        assert not res.ld32.cpu().numpy().any()
        return
    idx = res.cell_index(rows, cols)
    o = c_oracle.Panel(codes).triangle(libm_pow=False)
    ld32 = res.ld32.cpu().numpy()[idx]
    assert np.array_equal(res.n11.cpu().numpy().view(np.uint32)[idx], o["n11"][rows, cols])
    assert np.array_equal(k_of(ld32[:, 0]), np.rint(o["rsq_rnd"][rows, cols] * 1e4).astype(np.int64))
    assert np.array_equal(k_of(ld32[:, 1]), np.rint(o["dp_rnd"][rows, cols] * 1e4).astype(np.int64))
    assert np.array_equal(flags_of(ld32), o["flags"][rows, cols])
    raw = res.raw.cpu().numpy()[idx]
    assert np.array_equal(raw[:, 0], o["rsq_raw"][rows, cols]) and np.array_equal(raw[:, 1], o["dp_raw"][rows, cols])


def test_popcount_and_mfma_kernels_agree_bitwise(gpu):
    import torch

    from ld_tools_amd import PackedPanel, ld_triangle, ops, synth

    p = PackedPanel.from_codes(synth.synth_codes_device(1500, 5008, seed=21, miss=0.004))
    ops.set_triangle_path("popcount")
    a = ld_triangle(p, want_raw=True, want_n11=True)
    for mm in ("mfma", "fp4"):
        ops.set_triangle_path(mm)
        b = ld_triangle(p, want_raw=True, want_n11=True)
        ops.set_triangle_path("auto")
        assert torch.equal(a.n11, b.n11), mm
        assert torch.equal(a.ld32.view(torch.int32), b.ld32.view(torch.int32)), mm      # incl. the -0.0f int-0 marks
        assert torch.equal(a.raw.view(torch.int64), b.raw.view(torch.int64)), mm
        # ragged unit ranges (not multiples of 8 small units) on the matrix paths
        ops.set_triangle_path(mm)
        parts = [(0, 13), (13, 1001), (1001, p.n_units)]
        pieces = [ld_triangle(p, unit_range=r, want_n11=True) for r in parts]
        ops.set_triangle_path("auto")
        assert torch.equal(torch.cat([x.n11 for x in pieces]), a.n11), mm
        assert torch.equal(torch.cat([x.ld32 for x in pieces]).view(torch.int32), a.ld32.view(torch.int32)), mm


def test_too_many_haplotypes_is_an_error(gpu):
    from ld_tools_amd import LdxError, PackedPanel

    with pytest.raises(LdxError):
        PackedPanel.empty(4, 10241)


def test_triangle_bench_size_against_oracle_rows(gpu, path):
    """configs[1] (10 000 x 5008): every cell of the FP4 kernel's triangle against the C oracle (three bands of rows for the
    two comparison kernels), plus size-independent properties."""
    from ld_tools_amd import PackedPanel, ld_triangle, pair_counts, synth
    from oracle import c_oracle

    n, h = 10000, 5008
    codes_d = synth.synth_codes_device(n, h, seed=synth.BENCH_SEED)
    p = PackedPanel.from_codes(codes_d)
    res = ld_triangle(p, want_n11=True)
    codes = codes_d.cpu().numpy()
    o = c_oracle.Panel(codes)
    assert np.array_equal(p.alt_counts(), o.acnt)
    ld32 = res.ld32.cpu().numpy()
    n11 = res.n11.cpu().numpy().view(np.uint32)
    if path == "fp4":
        # the product's kernel: EVERY one of the 49 995 000 cells against the oracle (the full-size configs met
        # the oracle on row bands only; everything else was kernel against kernel).  ~3.5 s of oracle time per thread.
        bands = [(r, min(r + 125, n)) for r in range(0, n, 125)]
        assert oracle_rows_against_cells(o, res, ld32, n11, bands) == n * (n - 1) // 2
    else:   # the comparison kernels: three bands here, every cell against the FP4 kernel's in the tests below / bench.py
        assert oracle_rows_against_cells(o, res, ld32, n11, [(1, 40), (5000, 5024), (9990, 10000)]) > 200000
    # properties over ALL cells: total n11 mass equals sum_h C(k_h, 2), k_h = ALT count of haplotype column h
    colsum = (codes == 1).sum(axis=0).astype(np.int64)
    assert int(n11.astype(np.int64).sum()) == int((colsum * (colsum - 1) // 2).sum())
    # values in range, and a strip row agrees with the rectangular n11 kernel
    rng = np.random.RandomState(0)
    rr = rng.randint(1, n, size=200000)
    cc = (rng.random_sample(200000) * rr).astype(np.int64)
    valid = ld32[res.cell_index(rr, cc)]
    assert (valid[:, 0] >= 0).all() and (valid[:, 0] <= 1.0001).all() and (valid[:, 1] <= 1.0001).all()
    sub = PackedPanel.from_codes(codes[4096:4096 + 256])
    blk = pair_counts(sub, p).cpu().numpy().view(np.uint32)
    rows = np.repeat(np.arange(4096, 4352), 4096)
    cols = np.tile(np.arange(4096), 256)
    assert np.array_equal(n11[res.cell_index(rows, cols)], blk[:, :4096].ravel())
    # the product variants (no n11 output: on the FP4 path that is the fp32 epilogue tier with its fp64 / mirror fallbacks)
    # give the very same cells, in both formats
    import torch
    plain = ld_triangle(p)
    assert torch.equal(plain.ld32.view(torch.int32), res.ld32.view(torch.int32))
    k16 = ld_triangle(p, fmt="k16").k16.cpu().numpy().view(np.uint16).astype(np.int64)
    want = np.where(np.signbit(ld32), 0x8000, k_of(ld32))
    # the two formats order the cells of a unit differently (include/ldx.h): bring the ld32-order array into k16 order
    from ld_tools_amd._lib import cell_offset
    r8, c = np.meshgrid(np.arange(8), np.arange(128), indexing="ij")
    src = np.empty(1024, dtype=np.int64)
    src[cell_offset(r8, c, "k16").ravel()] = cell_offset(r8, c, "ld32").ravel()
    want = want.reshape(-1, 1024, 2)[:, src, :].reshape(-1, 2)
    assert np.array_equal(k16, want)


def _config_panel(n, h):
    from ld_tools_amd import PackedPanel, synth
    codes_d = synth.synth_codes_device(n, h, seed=synth.BENCH_SEED)
    return codes_d, PackedPanel.from_codes(codes_d)


def test_config4_triangle_50k_x_1008(gpu):
    """BASELINE configs[4]: ld_triangle 50 000 SNPs x 1008 haplotypes (EUR-size sub-panel) at full size.  The three kernels
    agree bit for bit on all 1.25e9 pairs (both cell formats), bands of rows match the C oracle, and the n11 mass equals
    sum_h C(k_h, 2)."""
    import torch
    from ld_tools_amd import ld_triangle
    from oracle import c_oracle

    n, h = 50000, 1008
    codes_d, p = _config_panel(n, h)
    ref = ld_triangle(p, want_n11=True, path="fp4")
    for other in ("popcount", "mfma"):
        b = ld_triangle(p, want_n11=True, path=other)
        assert torch.equal(ref.n11, b.n11), other
        assert torch.equal(ref.ld32.view(torch.int32), b.ld32.view(torch.int32)), other
        del b
    plain = ld_triangle(p, path="fp4")                       # the product variant: fp32 tier + fallbacks
    assert torch.equal(plain.ld32.view(torch.int32), ref.ld32.view(torch.int32))
    del plain
    k16 = ld_triangle(p, fmt="k16", path="fp4")
    k_ref = torch.where(torch.signbit(ref.ld32), torch.full_like(ref.ld32, 32768.0), torch.round(ref.ld32.double() * 1e4).float())
    # the two formats order the cells of a unit differently (include/ldx.h): bring the ld32-order tensor into k16 order
    from ld_tools_amd._lib import cell_offset
    r8, c = np.meshgrid(np.arange(8), np.arange(128), indexing="ij")
    src = np.empty(1024, dtype=np.int64)
    src[cell_offset(r8, c, "k16").ravel()] = cell_offset(r8, c, "ld32").ravel()
    k_ref = k_ref.view(-1, 1024, 2)[:, torch.from_numpy(src).to(k_ref.device), :].reshape(-1, 2)
    assert torch.equal(k16.k16.to(torch.int32) & 0xFFFF, k_ref.to(torch.int32))
    for other in ("popcount", "mfma"):
        b = ld_triangle(p, fmt="k16", path=other)
        assert torch.equal(b.k16, k16.k16), other
        del b
    del k16, k_ref
    # Soak: the same launch 60 times into a poisoned buffer.  38 000 passes per launch, two workgroups per CU: a barrier
    # that does not order the LDS pass ticket (csrc/ldx_common.h, block_sync) showed here as one miscomputed pass in about
    # one launch of twenty-five -- and nowhere at 10 000 x 5008, where a launch has 25 times fewer passes.
    want = ld_triangle(p, fmt="k16", path="popcount").k16.clone()
    got = ld_triangle(p, fmt="k16", path="fp4")
    for it in range(60):
        got.k16.view(torch.int16).fill_(-1)
        ld_triangle(p, fmt="k16", path="fp4", out=got)
        assert torch.equal(got.k16, want), f"launch {it} of the soak differs from the popcount kernel"
    del want, got
    codes = codes_d.cpu().numpy()
    o = c_oracle.Panel(codes)
    assert np.array_equal(p.alt_counts(), o.acnt) and np.array_equal(p.ref_counts(), o.rcnt)
    # 40 % of the rows, all their cells, against the oracle: 40 rows out of every 100 (20 000 rows, 5.0e8 cells) + both ends
    # (round 5: 40 of every 750)
    ld32_h = ref.ld32.cpu().numpy()
    n11_h = ref.n11.cpu().numpy().view(np.uint32)
    bands = [(1, 30), (49996, 50000)] + [(r, min(r + 40, n)) for r in range(30, n, 100)]
    assert sum(b[1] - b[0] for b in bands) >= 0.39 * n
    assert oracle_rows_against_cells(o, ref, ld32_h, n11_h, bands) > 4.8e8
    del ld32_h, n11_h
    colsum = (codes == 1).sum(axis=0).astype(np.int64)
    assert int(ref.n11.to(torch.int64).sum().item()) == int((colsum * (colsum - 1) // 2).sum())


def oracle_rows_against_k16(o, res, k16, bands, workers=8):
    """oracle_rows_against_cells for the PRODUCT variant's 4-byte cells (no n11 side output: the fp32 tier and its fallbacks):
    k of r^2 and D' and the int-0 marks of every cell (row > col) of the rows in `bands` against the C oracle."""
    from concurrent.futures import ThreadPoolExecutor

    def one(band):
        r0, r1 = band
        t = o.triangle_band(r0, r1, libm_pow=True, want=("rsq_rnd", "dp_rnd", "flags"))
        counts = np.arange(r0, r1, dtype=np.int64)
        rows = np.repeat(counts, counts)
        cols = np.concatenate([np.arange(i, dtype=np.int64) for i in range(r0, r1)])
        u = k16[res.cell_index(rows, cols)].astype(np.int64)
        rb = rows - r0
        fl = t["flags"][rb, cols]
        assert not (u == 0x7FFF).any(), ("escape cell", band)
        want_r = np.where((fl & 2) != 0, 0x8000, np.rint(t["rsq_rnd"][rb, cols] * 1e4).astype(np.int64))
        want_d = np.where((fl & 1) != 0, 0x8000, np.rint(t["dp_rnd"][rb, cols] * 1e4).astype(np.int64))
        assert np.array_equal(u[:, 0], want_r), ("r_square", band)
        assert np.array_equal(u[:, 1], want_d), ("d_prime", band)
        return len(rows)

    bands = [b for b in bands if b[1] > max(b[0], 1)]
    with ThreadPoolExecutor(max_workers=workers) as pool:
        return sum(pool.map(one, bands))


@pytest.mark.parametrize("n,h", [(10000, 5008), (50000, 1008)])
def test_panels_with_monomorphic_and_missing_code_snps_at_bench_sizes(gpu, n, h):
    """panels that are not all ordinary, at configs[1]'s and configs[4]'s sizes -- 30 % of the SNPs
    monomorphic (what a sub-panel of the ALL-panel variants holds) AND 0.1 % code 2 in 20 % of the rows.  The product variant
    of the FP4 kernel (fp32 tier with forced int-0 cells for the degenerate rows / columns, a few missing codes on the common
    path) against the popcount kernel, every cell, both formats; and >= 5 % of the rows, all their cells, against the C oracle."""
    import torch
    from ld_tools_amd import PackedPanel, ld_triangle, synth
    from oracle import c_oracle

    codes_d = synth.synth_codes_device(n, h, seed=synth.BENCH_SEED, miss=0.001, mono=0.3, miss_rows=0.2)
    p = PackedPanel.from_codes(codes_d)
    got = ld_triangle(p, fmt="k16", path="fp4")
    want = ld_triangle(p, fmt="k16", path="popcount")
    assert torch.equal(got.k16, want.k16)
    del want
    if n <= 10000:
        a, b = ld_triangle(p, fmt="ld32", path="fp4"), ld_triangle(p, fmt="ld32", path="popcount")
        assert torch.equal(a.ld32.view(torch.int32), b.ld32.view(torch.int32))
        del a, b
    codes = codes_d.cpu().numpy()
    flat = (codes == codes[:, :1]).all(axis=1)
    assert 0.25 * n < flat.sum() < 0.35 * n and 0.05 * n < (codes == 2).any(axis=1).sum() < 0.25 * n   # (at 1008 haplotypes a flagged row has a code 2 with probability 0.63)
    o = c_oracle.Panel(codes)
    k16 = got.k16.cpu().numpy().view(np.uint16)
    step = 500
    bands = [(1, 20), (n - 4, n)] + [(r, min(r + 27, n)) for r in range(20, n, step)]
    assert sum(b[1] - b[0] for b in bands) >= 0.05 * n
    assert oracle_rows_against_k16(o, got, k16, bands) >= 0.05 * n * (n - 1) / 2 * 0.9


def test_config2_area_100k_500kb(gpu):
    """BASELINE configs[2]: ld_area over a 100 000-SNP chromosome (positions 1 + 500 i: +-500 kb = +-1000 neighbours), every
    SNP a query, rounded r^2 >= 0.8.  The matrix-pipe band (FP4 and int8) and the popcount scan return the same hits;
    the hits of three query bands (first, middle, last 200 queries) equal the C oracle's window loop; the number of
    evaluated pairs equals the window populations."""
    from ld_tools_amd import ld_area, ops, synth
    from oracle import c_oracle

    n, h, flank = 100000, 5008, 500000
    codes_d, p = _config_panel(n, h)
    pos = synth.synth_positions(n, step=500)
    got = {}
    try:
        for path in ("fp4", "mfma", "popcount"):
            ops.set_area_path(path)
            hits = ld_area(p, pos, None, flank, "r_square", 0.8)
            got[path] = (hits.query.cpu().numpy(), hits.oppos.cpu().numpy(), hits.ld32.cpu().numpy().view(np.uint32), hits.n_pairs)
    finally:
        ops.set_area_path("auto")
    a = got["fp4"]
    for other in ("mfma", "popcount"):
        b = got[other]
        assert len(a[0]) == len(b[0]) and a[3] == b[3], other
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), other
    # window populations: max(0, pos - flank) < pos_o <= pos + flank (pysam's half-open fetch, ld_area.py:174-177): 999
    # neighbours below (the SNP exactly 500 kb below is outside), 1000 above, clipped at the chromosome ends
    idx = np.arange(n)
    assert a[3] == int((np.minimum(idx, 999) + np.minimum(n - 1 - idx, 1000)).sum())
    assert len(a[0]) > 100000                                   # the block-LD panel has plenty of r^2 >= 0.8 pairs
    codes = codes_d.cpu().numpy()
    o = c_oracle.Panel(codes)
    for q0 in (0, 49900, n - 200):
        qs = np.arange(q0, q0 + 200)
        hq, ho, hr, hd, hf = o.area(pos, qs, flank, 0, 0.8, libm_pow=True)
        m = (a[0] >= q0) & (a[0] < q0 + 200)
        assert np.array_equal(a[0][m], hq) and np.array_equal(a[1][m], ho), q0
        ld = a[2][m].view(np.float32).reshape(-1, 2)
        assert np.array_equal(k_of(ld[:, 0]), np.rint(hr * 1e4).astype(np.int64)), q0
        assert np.array_equal(k_of(ld[:, 1]), np.rint(hd * 1e4).astype(np.int64)), q0
        assert np.array_equal(flags_of(ld), hf), q0


def test_triangle_random_shapes_both_kernels_and_oracle(gpu):
    """Forty seeded random shapes (1..1400 SNPs, 1..2600 haplotypes, with and without missing codes and monomorphic
    rows): the three kernels agree bit for bit on every cell, and the FP4 kernel equals the C oracle on all of them."""
    import torch
    from ld_tools_amd import PackedPanel, ld_triangle, ops, synth
    from oracle import c_oracle

    rng = np.random.RandomState(2026)
    try:
        for case in range(40):
            n = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 500, 777, 1025, 1400]))
            h = int(rng.choice([1, 2, 31, 32, 33, 127, 128, 129, 500, 1008, 1023, 1025, 2600]))
            miss = float(rng.choice([0.0, 0.0, 0.003, 0.05]))
            codes = synth.synth_codes_host(n, h, seed=1000 + case, miss=miss)
            if n > 4 and rng.rand() < 0.5:
                codes[rng.randint(n)] = rng.randint(2)          # a monomorphic row
            p = PackedPanel.from_codes(codes)
            got = {}
            for path in ("fp4", "mfma", "popcount"):
                ops.set_triangle_path(path)
                r = ld_triangle(p, want_n11=True)
                got[path] = (r.ld32.clone().view(torch.int32), r.n11.clone(), r)
            for mm in ("mfma", "fp4"):
                assert torch.equal(got[mm][0], got["popcount"][0]) and torch.equal(got[mm][1], got["popcount"][1]), (n, h, miss, mm)
            if n >= 2:
                rows, cols = np.tril_indices(n, -1)
                res = got["fp4"][2]
                idx = res.cell_index(rows, cols)
                o = c_oracle.Panel(codes).triangle(libm_pow=True, want=("n11", "rsq_rnd", "dp_rnd", "flags"))
                ld32 = res.ld32.cpu().numpy()[idx]
                assert np.array_equal(res.n11.cpu().numpy().view(np.uint32)[idx], o["n11"][rows, cols]), (n, h, miss)
                ok = (o["rsq_rnd"][rows, cols] < 1000) & (o["dp_rnd"][rows, cols] < 1000)
                assert np.array_equal(k_of(ld32[ok, 0]), np.rint(o["rsq_rnd"][rows, cols][ok] * 1e4).astype(np.int64)), (n, h, miss)
                assert np.array_equal(k_of(ld32[ok, 1]), np.rint(o["dp_rnd"][rows, cols][ok] * 1e4).astype(np.int64)), (n, h, miss)
                assert np.array_equal(flags_of(ld32), o["flags"][rows, cols]), (n, h, miss)
    finally:
        ops.set_triangle_path("auto")


def test_triangle_half_height_tickets_agree(gpu, path):
    """Whole passes, all passes halved (two 32-row tickets each) and a mix give identical results
    (ldx_debug_force_short_passes forces the number of halved passes; by default only tiny launches and the last quarter
    round of large ones are halved)."""
    import torch
    from ld_tools_amd import PackedPanel, ld_triangle, synth
    from ld_tools_amd._lib import lib

    if path == "popcount":
        pytest.skip("tickets are a matrix-kernel matter")
    for n, h, miss in [(3000, 5008, 0.0), (1111, 777, 0.01)]:
        p = PackedPanel.from_codes(synth.synth_codes_device(n, h, seed=9, miss=miss))
        got = []
        try:
            for short in (0, 1000000, 37):
                lib.ldx_debug_force_short_passes(short)
                r = ld_triangle(p, want_n11=True)
                got.append((r.ld32.clone().view(torch.int32), r.n11.clone()))
        finally:
            lib.ldx_debug_force_short_passes(-1)
        for a, b in got[1:]:
            assert torch.equal(a, got[0][0]) and torch.equal(b, got[0][1])


def test_fp32_tier_parks_the_rows_and_columns_of_odd_snps(gpu):
    """Round 4: a SNP that is not ordinary parks its own row / column of lane-steps in the fp32 tier instead of sending the whole
    unit to the fp64 epilogue; round 6: monomorphic ALT / REF and all-missing SNPs do not even park (their cells are forced to the
    int-0 code on the common path) and SNPs with a few missing codes are ordinary; many missing codes still park.  Panels with one, a few
    and MANY such SNPs per tile (the last overflows the per-wave queue: the unit is redone whole) through the product
    variant of the FP4 kernel -- interior units only exist from ~400 SNPs on -- against the popcount kernel, both cell
    formats, and a band of rows against the C oracle."""
    import torch
    from ld_tools_amd import PackedPanel, ld_triangle, synth
    from oracle import c_oracle

    n, h = 1500, 1008
    rng = np.random.RandomState(11)
    for n_odd in (1, 6, 40, 400):
        codes = synth.synth_codes_host(n, h, seed=5 + n_odd, miss=0.0)
        rows = rng.choice(n, size=n_odd, replace=False)
        for k, r in enumerate(rows):
            kind = k % 6
            if kind == 0:
                codes[r] = 0                                   # monomorphic REF   (degenerate: forced cells, round 6)
            elif kind == 1:
                codes[r] = 1                                   # monomorphic ALT   (degenerate)
            elif kind == 2:
                codes[r, ::7] = 2                              # many missing codes: a + r < n   (odd: parks)
            elif kind == 3:
                codes[r] = 2                                   # nothing but missing codes   (a == 0: degenerate)
            elif kind == 4:
                codes[r, 5:9] = 2                              # a few missing codes   (ordinary since round 6)
            else:
                codes[r] = 1
                codes[r, ::3] = 2                              # no REF allele, a < n   (degenerate too: int 0 whatever n11 is)
        p = PackedPanel.from_codes(codes)
        for fmt in ("k16", "ld32"):
            got = ld_triangle(p, fmt=fmt, path="fp4")          # no side outputs: the fp32 tier + its fallbacks
            want = ld_triangle(p, fmt=fmt, path="popcount")
            view = torch.int16 if fmt == "k16" else torch.int32
            assert torch.equal(got.cells.view(view), want.cells.view(view)), (n_odd, fmt)
        o = c_oracle.Panel(codes)
        r0 = int(rows[0]) if rows[0] > 0 else 1
        t = o.triangle(r0, r0 + 1, libm_pow=True)
        cols = np.arange(r0, dtype=np.int64)
        res = ld_triangle(p, fmt="k16", path="fp4")
        kk, int0, esc = res.k_and_int0(res.cell_index(np.full(r0, r0), cols))
        want_k = np.stack([np.rint(t["rsq_rnd"][r0, :r0] * 1e4), np.rint(t["dp_rnd"][r0, :r0] * 1e4)], axis=1)
        ok = ~esc
        assert np.array_equal(kk[ok], want_k[ok].astype(np.int64)), n_odd
        assert np.array_equal(int0[:, 0], (t["flags"][r0, :r0] & 2) != 0) and np.array_equal(int0[:, 1], (t["flags"][r0, :r0] & 1) != 0)


def test_triangle_on_many_streams(gpu):
    """Many streams, used one after another and two at a time, each launch with its own result buffer and therefore its own
    pass-scheduler workspace (TriangleResult.ws): the single-stream result every time."""
    import torch
    from ld_tools_amd import PackedPanel, ld_triangle, synth

    p = PackedPanel.from_codes(synth.synth_codes_device(300, 1008, seed=2))
    ref = ld_triangle(p).ld32.clone()
    torch.cuda.synchronize()
    for k in range(300):
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(s1):
            o1 = ld_triangle(p)
        with torch.cuda.stream(s2):
            o2 = ld_triangle(p)
        s1.synchronize()
        s2.synchronize()
        assert torch.equal(o1.ld32.view(torch.int32), ref.view(torch.int32)), k
        assert torch.equal(o2.ld32.view(torch.int32), ref.view(torch.int32)), k


def test_graph_of_launches_into_alternating_buffers(gpu):
    """Round 5: consecutive kernel nodes of a HIP graph that share no written buffer argument are chained WITHOUT the cache
    write-back a stream gives between two launches; the ticket counters, re-armed with plain stores, then read as exhausted
    to the next node's atomics, which computed a third of its triangle (tools/gpu_streams_dbg.py; the sequence below is the
    one that showed it: the one-stream graph captured BEFORE side streams are used and a fork / join graph is captured).
    The counters are now only ever written with agent-scope atomics (csrc/ldx_common.h, store_agent)."""
    import torch
    from ld_tools_amd import PackedPanel, ld_triangle, synth

    p = PackedPanel.from_codes(synth.synth_codes_device(6000, 5008, seed=4))     # 588 passes: more than the 512 static tickets
    ref = ld_triangle(p, fmt="k16")
    outs = [ld_triangle(p, fmt="k16"), ld_triangle(p, fmt="k16")]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()

    def one(count):
        for k in range(count):
            ld_triangle(p, out=outs[k & 1], fmt="k16")

    def two(count):
        cur = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(cur)
        for k in range(count):
            with torch.cuda.stream(streams[k & 1]):
                ld_triangle(p, out=outs[k & 1], fmt="k16")
        for st in streams:
            cur.wait_stream(st)

    graphs = []
    for fn in (one, two):
        fn(4)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn(6)
        g.replay()
        torch.cuda.synchronize()
        graphs.append(g)
    for rnd in range(3):
        for name, g in zip(("one stream", "two streams"), graphs):
            for o in outs:
                o.cells.fill_(-1)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            for k, o in enumerate(outs):
                assert torch.equal(o.cells, ref.cells), (rnd, name, k)


def test_two_triangle_graphs_of_one_capture_stream_replayed_at_once(gpu):
    """A launch recorded under stream capture gets its own ticket counters (round 5; csrc/ldx_mfma.hip, g_capt): torch captures
    every graph on one and the same side stream, so two graphs -- two panels here, several launches each -- used to carry
    the counters of that stream's slot, and replayed at the same time on two streams they would have shared them.  Two
    threads replay one graph each, on their own streams, fifty times; every result buffer equals the eager result."""
    import threading

    import torch
    from ld_tools_amd import PackedPanel, ld_triangle, synth

    jobs = []
    for n, seed in ((6000, 21), (7000, 22)):                     # more passes than the 512 static tickets: the counters are in use
        p = PackedPanel.from_codes(synth.synth_codes_device(n, 1008, seed=seed))
        ref = ld_triangle(p, fmt="k16")
        outs = [ld_triangle(p, fmt="k16") for _ in range(2)]
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(4):
                ld_triangle(p, out=outs[k & 1], fmt="k16")
        jobs.append((g, outs, ref))
    errors = []

    def worker(g, outs, ref):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for rep in range(50):
                    for o in outs:
                        o.cells.fill_(-1)
                    g.replay()
                    st.synchronize()
                    if not all(torch.equal(o.cells, ref.cells) for o in outs):
                        errors.append(rep)
                        return
        except Exception as exc:   # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_area_plans_replayed_on_two_streams_at_once(gpu):
    """every ld_area plan graph used to carry the ticket-counter slot of torch's one capture stream, so two plans
    replayed at the same time on two streams shared their counters.  The band's counters now live in the plan's own
    workspace: two panels' plans, replayed concurrently from two threads on two streams, keep returning the hits of the
    popcount scan."""
    import threading

    import torch
    from ld_tools_amd import PackedPanel, ld_area, ops, synth

    cases = []
    for n, seed in ((30000, 11), (24000, 12)):
        p = PackedPanel.from_codes(synth.synth_codes_device(n, 1008, seed=seed))
        pos = torch.as_tensor(synth.synth_positions(n, step=500)).to(p.device)
        ops.set_area_path("popcount")
        want = ld_area(p, pos, None, 100000, "r_square", 0.8, use_graph=False)
        ops.set_area_path("auto")
        for _ in range(2):                                   # the second repetition captures the plan's graph
            got = ld_area(p, pos, None, 100000, "r_square", 0.8, check_positions=False)
        assert [pl for k, pl in p._area_plans.items() if k != "all_rows"][0].graph
        cases.append((p, pos, want, got))
    torch.cuda.synchronize()
    errors = []

    def worker(p, pos, want):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for rep in range(40):
                    got = ld_area(p, pos, None, 100000, "r_square", 0.8, check_positions=False)
                    ok = len(got) == len(want) and torch.equal(got.query, want.query) and torch.equal(got.oppos, want.oppos) \
                        and torch.equal(got.ld32.view(torch.int32), want.ld32.view(torch.int32))
                    if not ok:
                        errors.append((p.n_snps, rep, len(got), len(want)))
                        return
        except Exception as exc:   # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(p, pos, want)) for p, pos, want, _ in cases]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for p, *_ in cases:
        p.clear_area_plans()
        assert "_area_plans" not in p.__dict__


def test_area_repeated_queries_are_one_query(gpu, area_path):
    """a query list with repetitions whose LENGTH equals the number of SNPs must not be mistaken for "every SNP is
    a query" (the band drops its query mask then); ops.ld_area makes the list strictly ascending (include/ldx.h)."""
    import torch
    from ld_tools_amd import PackedPanel, ld_area, synth

    n = 2000
    p = PackedPanel.from_codes(synth.synth_codes_device(n, 1008, seed=9))
    pos = torch.as_tensor(synth.synth_positions(n, step=500)).to(p.device)
    uniq = list(range(0, n, 2))
    dup = uniq + uniq                                        # n entries, half of the SNPs
    a = ld_area(p, pos, uniq, 50000, "r_square", 0.3)
    b = ld_area(p, pos, dup, 50000, "r_square", 0.3)
    assert len(a) == len(b) > 0 and torch.equal(a.query, b.query) and torch.equal(a.oppos, b.oppos)
    assert torch.equal(a.ld32.view(torch.int32), b.ld32.view(torch.int32))
    assert bool((a.query % 2 == 0).all())


def test_triangle_on_a_thousand_raw_streams(gpu):
    """a driver that creates a raw hipStream_t per chromosome / table.  Rounds 3-5 kept the
    matrix kernel's ticket counters in a pool of 256 (device, stream) slots inside the library; since round 6 they live in the
    CALLER's workspace (include/ldx.h, ldx_triangle_ex_dev), so there is nothing to run out of: 1 000 streams made with
    hipStreamCreate (ctypes on libamdhip64: torch recycles a pool of 32) are used once each with an EXPLICIT matrix-pipe path
    and destroyed -- one shared workspace, because the launches follow one another -- then 300 streams are kept alive and busy
    at once, a workspace each; then recycled stream handles with launches still queued (the old pool could
    hand one slot to two live launches there).  Every result equals the single-stream one; no launch is ever refused."""
    import ctypes
    import torch
    from ld_tools_amd import PackedPanel, _lib, ld_triangle, synth
    from ld_tools_amd._lib import lib

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    p = PackedPanel.from_codes(synth.synth_codes_device(300, 1008, seed=2))
    ref = ld_triangle(p, fmt="k16", path="fp4")
    torch.cuda.synchronize()
    cells = ref.k16.numel() // 2
    out = torch.empty_like(ref.k16)
    ws_bytes = lib.ldx_triangle_workspace_bytes()

    def workspace():
        return torch.zeros(ws_bytes, dtype=torch.uint8, device=p.device)

    def launch(stream, path, dst, ws, panel=p):
        return lib.ldx_triangle_ex_dev(panel.alt.data_ptr(), panel.fa.data_ptr(), panel.fr.data_ptr(), panel.q.data_ptr(),
                                       panel.n_snps, panel.n_hap, 0, panel.n_units, path, _lib.FORMATS["k16"], dst.data_ptr(),
                                       None, None, ws.data_ptr(), ws_bytes, stream)

    shared = torch.full((ws_bytes,), 0x5A, dtype=torch.uint8, device=p.device)     # garbage, then the library's own initialiser
    assert lib.ldx_triangle_workspace_init_dev(shared.data_ptr(), ws_bytes, torch.cuda.current_stream().cuda_stream) == 0
    assert lib.ldx_triangle_workspace_init_dev(shared.data_ptr(), ws_bytes - 1, None) < 0           # too small: refused
    assert lib.ldx_triangle_ex_dev(p.alt.data_ptr(), p.fa.data_ptr(), p.fr.data_ptr(), p.q.data_ptr(), p.n_snps, p.n_hap, 0, p.n_units,
                                   3, _lib.FORMATS["k16"], out.data_ptr(), None, None, shared.data_ptr() + 8, ws_bytes, None) < 0   # misaligned
    torch.cuda.synchronize()
    for k in range(1000):
        st = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(st)) == 0
        out.fill_(-1)
        torch.cuda.synchronize()
        rc = launch(st, 3, out, shared)              # LDX_PATH_FP4, explicit: no popcount fallback may hide a failure
        assert rc == 0, (k, rc, lib.ldx_last_error())
        assert hip.hipStreamSynchronize(st) == 0
        assert torch.equal(out, ref.k16), k
        assert hip.hipStreamDestroy(st) == 0
    assert out.numel() == 2 * cells
    assert int(shared.view(torch.int32)[:8].abs().sum().item()) == 0      # every launch left the counters re-armed
    # 300 live streams, a long matrix-pipe kernel in flight on each (the old pool had 256 slots)
    big = PackedPanel.from_codes(synth.synth_codes_device(6000, 1008, seed=3))
    bref = ld_triangle(big, fmt="k16", path="popcount")
    torch.cuda.synchronize()
    outs, streams, spaces = [], [], []
    for k in range(300):
        st = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(st)) == 0
        streams.append(st)
        o = torch.empty_like(bref.k16)
        spaces.append(workspace())
        rc = launch(st, 3 if k & 1 else 0, o, spaces[-1], big)       # explicit FP4 / AUTO alternately: never refused
        assert rc == 0, (k, rc, lib.ldx_last_error())
        outs.append(o)
    for st in streams:
        assert hip.hipStreamSynchronize(st) == 0
    for k, o in enumerate(outs):
        assert torch.equal(o, bref.k16), k
    for st in streams:
        assert hip.hipStreamDestroy(st) == 0
    # recycled handles: a stream is destroyed with its launch still queued behind a long kernel, the next stream created may
    # get its handle and launches at once on its own workspace; both results must be right (nothing is keyed by the handle)
    for rnd in range(6):
        sx, sy = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(sx)) == 0
        blocker = torch.empty_like(bref.k16)
        ox, oy = torch.full_like(ref.k16, -1), torch.full_like(ref.k16, -1)
        wx, wy, wb = workspace(), workspace(), workspace()
        torch.cuda.synchronize()
        assert launch(sx, 1, blocker, wb, big) == 0              # POPCOUNT: ~ms
        assert launch(sx, 3, ox, wx) == 0, lib.ldx_last_error()  # queued behind the blocker
        assert hip.hipStreamDestroy(sx) == 0                     # destroyed with work in flight (HIP finishes it)
        assert hip.hipStreamCreate(ctypes.byref(sy)) == 0        # may well be sx's handle again
        assert launch(sy, 3, oy, wy) == 0, lib.ldx_last_error()
        assert hip.hipStreamSynchronize(sy) == 0
        torch.cuda.synchronize()
        assert torch.equal(ox, ref.k16) and torch.equal(oy, ref.k16), rnd
        assert hip.hipStreamDestroy(sy) == 0


def test_triangle_without_a_workspace_deals_passes_round_robin(gpu):
    """include/ldx.h: workspace = NULL is allowed -- the passes are dealt round-robin, no counter anywhere -- and gives the same
    cells (ldx_triangle_dev, the plain entry point, runs that way).  Panels of less than one round, a few rounds and with
    halved passes; both matrix kernels; launches on two streams at once, which would share nothing."""
    import torch
    from ld_tools_amd import PackedPanel, _lib, ld_triangle, synth
    from ld_tools_amd._lib import lib

    for n, h, seed in ((700, 1008, 5), (9000, 1008, 6), (5000, 5008, 7)):
        p = PackedPanel.from_codes(synth.synth_codes_device(n, h, seed=seed))
        ref = ld_triangle(p, fmt="k16", path="popcount")
        ref32 = ld_triangle(p, fmt="ld32", path="popcount")
        torch.cuda.synchronize()
        for path in (3, 2, 0):
            outs = [torch.full_like(ref.k16, -1) for _ in range(2)]
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            torch.cuda.synchronize()                         # the fills run on torch's current stream, the launches on two others
            for o, st in zip(outs, streams):
                rc = lib.ldx_triangle_ex_dev(p.alt.data_ptr(), p.fa.data_ptr(), p.fr.data_ptr(), p.q.data_ptr(), p.n_snps, p.n_hap,
                                             0, p.n_units, path, _lib.FORMATS["k16"], o.data_ptr(), None, None, None, 0,
                                             st.cuda_stream)
                assert rc == 0, lib.ldx_last_error()
            torch.cuda.synchronize()
            for o in outs:
                assert torch.equal(o, ref.k16), (n, h, path)
        o32 = torch.full_like(ref32.ld32, float("nan"))
        rc = lib.ldx_triangle_dev(p.alt.data_ptr(), p.fa.data_ptr(), p.fr.data_ptr(), p.q.data_ptr(), p.n_snps, p.n_hap, 0,
                                  p.n_units, o32.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.ldx_last_error()
        torch.cuda.synchronize()
        assert torch.equal(o32.view(torch.int32), ref32.ld32.view(torch.int32)), (n, h)


def test_seventy_thousand_captured_triangle_launches(gpu):
    """rounds 5's captured launches each consumed a private counter set for good -- 65 536 per process and
    device, after which LDX_PATH_AUTO silently ran the 5x slower popcount kernel.  70 graphs of 1 000 launches each are
    captured here (and dropped again) with an EXPLICIT FP4 path: none is refused, and the last graph still computes the
    triangle.  The workspace is the result's own (TriangleResult.ws)."""
    import torch
    from ld_tools_amd import PackedPanel, ld_triangle, synth

    p = PackedPanel.from_codes(synth.synth_codes_device(200, 256, seed=12))
    ref = ld_triangle(p, fmt="k16", path="popcount")
    out = ld_triangle(p, fmt="k16", path="fp4")
    torch.cuda.synchronize()
    g = None
    for k in range(70):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(1000):
                ld_triangle(p, out=out, fmt="k16", path="fp4")
        if k % 23 == 0 or k == 69:
            out.cells.fill_(-1)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out.cells, ref.cells), k
    del g


def test_first_triangle_launch_of_a_process_under_capture(gpu):
    """the first matrix-pipe launch of a process used to allocate pinned host memory (the old slot pool's
    finished-launch words), which HIP refuses on a capturing thread.  Nothing is allocated at launch any more; a fresh process
    whose FIRST ld_triangle call of the panel's shape is recorded under capture gets the right triangle."""
    import subprocess

    code = (
        "import torch\n"
        "from ld_tools_amd import PackedPanel, ld_triangle, synth\n"
        "from ld_tools_amd import ops\n"
        "from ld_tools_amd._lib import lib\n"
        "p = PackedPanel.from_codes(synth.synth_codes_device(5000, 1008, seed=3))\n"
        "cells = p.n_units * 1024\n"
        "out = ops.TriangleResult(p.n_snps, 0, p.n_units, k16=torch.full((cells, 2), -1, dtype=torch.int16, device=p.device),\n"
        "                         ws=torch.zeros(lib.ldx_triangle_workspace_bytes(), dtype=torch.uint8, device=p.device))\n"
        "torch.cuda.synchronize()\n"
        "g = torch.cuda.CUDAGraph()\n"
        "with torch.cuda.graph(g):\n"
        "    ld_triangle(p, out=out, fmt='k16', path='fp4')\n"
        "g.replay(); torch.cuda.synchronize()\n"
        "ref = ld_triangle(p, fmt='k16', path='popcount'); torch.cuda.synchronize()\n"
        "assert torch.equal(out.cells, ref.cells)\n"
        "print('CAPTURE_FIRST_OK')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0 and "CAPTURE_FIRST_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_triangle_100k_shard_of_eight(gpu):
    """configs[3] (100 000 x 5008 over 8 GPUs): the unit range rank 3 of 8 would own, on one card.  The two kernels
    agree bit for bit on all 6.2e8 pairs of the shard, rows inside it match the C oracle, and the n11 mass of the
    shard equals what the rectangular-block kernel counts for the same cells."""
    import torch
    from ld_tools_amd import PackedPanel, dist, ld_triangle, ops, synth
    from ld_tools_amd._lib import UNIT_PAIRS, cell_offset
    from oracle import c_oracle

    n, h = 100000, 5008
    codes_d = synth.synth_codes_device(n, h, seed=synth.BENCH_SEED)
    p = PackedPanel.from_codes(codes_d)
    u0, u1 = dist.unit_partition(n, 8)[3]
    try:
        ops.set_triangle_path("fp4")
        a = ld_triangle(p, unit_range=(u0, u1), want_n11=True)
        for other in ("popcount", "mfma"):
            ops.set_triangle_path(other)
            b = ld_triangle(p, unit_range=(u0, u1), want_n11=True)
            assert torch.equal(a.n11, b.n11), other
            assert torch.equal(a.ld32.view(torch.int32), b.ld32.view(torch.int32)), other
            del b
    finally:
        ops.set_triangle_path("auto")
    assert a.ld32.shape[0] == (u1 - u0) * UNIT_PAIRS
    # cells of a few rows that lie inside the shard, against the oracle
    codes = codes_d.cpu().numpy()
    o = c_oracle.Panel(codes)
    npad = ((n + 127) // 128) * 128
    G = npad // 8
    checked = 0
    for row in (61007, 61008, 70001, 99999):
        cols = np.arange(row, dtype=np.int64)
        t_ = cols // 128
        u = t_ * G - 8 * t_ * (t_ - 1) + (row // 8 - 16 * t_)
        m = (u >= u0) & (u < u1)
        if not m.any():
            continue
        idx = (u[m] - u0) * UNIT_PAIRS + cell_offset(row % 8, cols[m] % 128, "ld32")
        c = cols[m]
        want_n = o.pair_counts(row, row + 1, int(c[0]), int(c[-1]) + 1)[0]        # the columns of a row's units are contiguous
        assert len(want_n) == len(c)
        k = len(c)
        _, _, w_rsq, w_dp, w_flags = c_oracle.ld_from_counts_v(
            h, want_n, np.full(k, o.acnt[row], np.uint32), np.full(k, o.rcnt[row], np.uint32), o.acnt[c], o.rcnt[c],
            libm_pow=True)                                                            # var_1 = row, var_2 = column
        got = a.ld32[torch.from_numpy(idx).to(a.ld32.device)].cpu().numpy()
        got_n = a.n11[torch.from_numpy(idx).to(a.ld32.device)].cpu().numpy().view(np.uint32)
        assert np.array_equal(got_n, want_n)
        assert np.array_equal(k_of(got[:, 0]), np.rint(w_rsq * 1e4).astype(np.int64))
        assert np.array_equal(k_of(got[:, 1]), np.rint(w_dp * 1e4).astype(np.int64))
        assert np.array_equal(flags_of(got), w_flags)
        checked += k
    assert checked > 20000


def test_config3_triangle_100k_all_eight_shards(gpu):
    """BASELINE configs[3] at FULL size on one card: 100 000 SNPs x 5008 haplotypes, all eight unit ranges of
    dist.unit_partition(100000, 8) one after the other (4 B cells + counts: 5 GB per shard, 5.0e9 pairs in all) -- what the
    eight ranks compute, minus the RCCL hop (tests/test_gpu_dist.py, tests/test_dist_gloo.py).  Per shard: the FP4 kernel
    and the popcount kernel agree on every cell and every count, the int8 kernel agrees on a sub-range of units, and 6.25 % of
    the rows -- all their cells, 3.1e8 -- match the C oracle; over the eight shards every unit is covered once and the n11
    mass of the whole triangle equals sum_h C(k_h, 2)."""
    import torch
    from ld_tools_amd import dist, ld_triangle, synth
    from ld_tools_amd._lib import UNIT_PAIRS, cell_offset
    from oracle import c_oracle

    n, h = 100000, 5008
    codes_d, p = _config_panel(n, h)
    parts = dist.unit_partition(n, 8)
    assert parts[0][0] == 0 and all(parts[k][1] == parts[k + 1][0] for k in range(7))
    codes = codes_d.cpu().numpy()
    o = c_oracle.Panel(codes)
    npad = ((n + 127) // 128) * 128
    G = npad // 8
    # 6.25 % of the rows, ALL their cells (3.1e8 of the 5.0e9), against the oracle -- every 16th row plus a few odd ones
    # (nine rows; round 5: every 100th) -- each row cut into the pieces that fall into the eight unit ranges
    from concurrent.futures import ThreadPoolExecutor

    rows_to_check = sorted(set(range(37, n, 16)) | {130, 20011, 38000, 52001, 61007, 70001, 84444, 93000, 99999})
    mass, checked, shards_checked = 0, 0, set()
    for r, (u0, u1) in enumerate(parts):
        a = ld_triangle(p, unit_range=(u0, u1), want_n11=True, fmt="k16", path="fp4")
        assert a.k16.shape[0] == (u1 - u0) * UNIT_PAIRS
        b = ld_triangle(p, unit_range=(u0, u1), want_n11=True, fmt="k16", path="popcount")
        assert torch.equal(a.n11, b.n11), f"shard {r}: counts, FP4 vs popcount"
        assert torch.equal(a.k16, b.k16), f"shard {r}: cells, FP4 vs popcount"
        del b
        plain = ld_triangle(p, unit_range=(u0, u1), fmt="k16", path="fp4")        # the product variant (fp32 tier + fallbacks)
        assert torch.equal(plain.k16, a.k16), f"shard {r}: product variant"
        del plain
        s0 = u0 + (u1 - u0) // 3                                                  # the third kernel on 4000 units of the shard
        c = ld_triangle(p, unit_range=(s0, s0 + 4000), fmt="k16", path="mfma")
        lo = (s0 - u0) * UNIT_PAIRS
        assert torch.equal(c.k16, a.k16[lo: lo + 4000 * UNIT_PAIRS]), f"shard {r}: int8 kernel"
        del c
        mass += int(a.n11.to(torch.int64).sum().item())
        pieces, idx_all = [], []
        for row in rows_to_check:
            cols = np.arange(row, dtype=np.int64)
            t_ = cols // 128
            u = t_ * G - 8 * t_ * (t_ - 1) + (row // 8 - 16 * t_)
            m = (u >= u0) & (u < u1)
            if not m.any():
                continue
            cm = cols[m]
            assert np.array_equal(cm, np.arange(cm[0], cm[-1] + 1))               # a row's units inside a range are contiguous
            idx_all.append((u[m] - u0) * UNIT_PAIRS + cell_offset(row % 8, cm % 128, "k16"))
            pieces.append((row, int(cm[0]), int(cm[-1]) + 1))
        if pieces:
            idx = torch.from_numpy(np.concatenate(idx_all)).to(a.k16.device)      # one gather per shard
            got_n = a.n11[idx].cpu().numpy().view(np.uint32)
            got_c = a.k16[idx].cpu().numpy().astype(np.int64) & 0xFFFF
            offs = np.concatenate([[0], np.cumsum([c1 - c0 for _, c0, c1 in pieces])])

            def one(k):
                row, c0, c1 = pieces[k]
                want_n = o.pair_counts(row, row + 1, c0, c1)[0]
                m_ = c1 - c0
                cm = np.arange(c0, c1)
                _, _, w_rsq, w_dp, w_flags = c_oracle.ld_from_counts_v(
                    h, want_n, np.full(m_, o.acnt[row], np.uint32), np.full(m_, o.rcnt[row], np.uint32), o.acnt[cm], o.rcnt[cm],
                    libm_pow=True)                                                # var_1 = row, var_2 = column
                sl = slice(offs[k], offs[k + 1])
                assert np.array_equal(got_n[sl], want_n), (r, row, "n11")
                int0 = np.stack([(w_flags & 2) != 0, (w_flags & 1) != 0], axis=1)
                want_k = np.stack([np.rint(w_rsq * 1e4), np.rint(w_dp * 1e4)], axis=1).astype(np.int64)
                assert np.array_equal(got_c[sl], np.where(int0, 0x8000, want_k)), (r, row)
                return m_

            with ThreadPoolExecutor(max_workers=8) as pool:
                checked += sum(pool.map(one, range(len(pieces))))
            shards_checked.add(r)
        del a
        torch.cuda.empty_cache()
    assert parts[-1][1] * UNIT_PAIRS >= n * (n - 1) // 2
    assert checked == sum(rows_to_check) and len(shards_checked) == 8      # every cell of those rows, in every shard
    colsum = (codes_d == 1).sum(dim=0, dtype=torch.int64)
    assert mass == int((colsum * (colsum - 1) // 2).sum().item())


# ------------------------------------------------------------------ ld_area
def test_area_matches_golden_drivers(gpu, area_path, drivers, panel_codes):
    from ld_tools_amd import PackedPanel, ld_area

    p = PackedPanel.from_codes(panel_codes[drivers["panel"]])
    f4 = p.alt_freq4().cpu().numpy()
    pos = drivers["positions"]
    for case in drivers["area"]:
        hits = ld_area(p, pos, case["queries"], case["flank"], case["measure"], case["thres"])
        want = sorted((tuple(h) for h in case["hits"]), key=lambda h: (h[0], h[1]))
        assert len(hits) == len(want), case["flank"]
        q = hits.query.cpu().numpy()
        o = hits.oppos.cpu().numpy()
        ld = hits.ld32.cpu().numpy()
        for k, w in enumerate(want):
            assert (q[k], o[k]) == (w[0], w[1])
            assert f4[o[k]] == w[2] and pos[o[k]] - pos[q[k]] == w[5]
            for col, wv in ((0, w[3]), (1, w[4])):
                g = ld[k, col]
                if isinstance(wv, int):
                    assert g == 0 and np.signbit(g)
                else:
                    assert not (g == 0 and np.signbit(g)) and round(float(g), 4) == wv


def test_area_banded_against_oracle(gpu, area_path):
    from ld_tools_amd import PackedPanel, ld_area, synth
    from oracle import c_oracle

    n, h = 3000, 1008
    codes = synth.synth_codes_host(n, h, seed=8, miss=0.002)
    pos = synth.synth_positions(n, step=500)
    p = PackedPanel.from_codes(codes)
    o = c_oracle.Panel(codes)
    # the last two: LONG hit lists per query (a threshold of 0 keeps the whole window: 1600 hits, then all 2999 = more than
    # one key tile of the workgroup rank sort that orders lists of more than 32 hits, csrc/ldx_area.hip)
    for (queries, flank, measure, thres) in [(None, 20000, "r_square", 0.8), (list(range(0, n, 7)), 150000, "d_prime", 1.0),
                                             (None, 3000, "r_square", 0.0), (None, 400000, "r_square", 0.0),
                                             (list(range(0, n, 50)), 10 ** 7, "d_prime", 0.0)]:
        hits = ld_area(p, pos, queries, flank, measure, thres)
        qs = np.arange(n) if queries is None else np.array(queries)
        hq, ho, hr, hd, hf = o.area(pos, qs, flank, 0 if measure == "r_square" else 1, thres, libm_pow=True)
        assert len(hits) == len(hq)
        assert np.array_equal(hits.query.cpu().numpy(), hq) and np.array_equal(hits.oppos.cpu().numpy(), ho)
        ld = hits.ld32.cpu().numpy()
        assert np.array_equal(k_of(ld[:, 0]), np.rint(hr * 1e4).astype(np.int64))
        assert np.array_equal(k_of(ld[:, 1]), np.rint(hd * 1e4).astype(np.int64))
        assert np.array_equal(flags_of(ld), hf)
        # pairs evaluated = window populations
        lo = np.searchsorted(pos, np.maximum(pos[qs] - flank, 0), side="right")
        hi = np.searchsorted(pos, pos[qs] + flank, side="right")
        assert hits.n_pairs == int((hi - lo).sum() - (flank > 0) * len(qs))


def test_area_hits_beyond_the_float_cell(gpu, area_path):
    """A (query, opposing) pair whose r^2 / D' exceed 1024 (missing codes, vanishing bound) is a hit for every threshold
    and comes back with its exact value -- 4080.4507 / 4948.0 for the reference's own known-answer tuple."""
    from ld_tools_amd import PackedPanel, ld_area
    from oracle import c_oracle

    codes = _panel_with_huge_values()
    p = PackedPanel.from_codes(codes)
    pos = 100 + 10 * np.arange(p.n_snps)
    for measure, thres in (("r_square", 0.8), ("d_prime", 0.99), ("r_square", 0.0)):
        hits = ld_area(p, pos, None, 10 ** 6, measure, thres)
        hq, ho, hr, hd, hf = c_oracle.Panel(codes).area(pos, np.arange(p.n_snps), 10 ** 6, 0 if measure == "r_square" else 1,
                                                         thres, libm_pow=True)
        assert np.array_equal(hits.query.cpu().numpy(), hq) and np.array_equal(hits.oppos.cpu().numpy(), ho)
        vals = hits.python_values(p)
        want = [(0 if f & 2 else r, 0 if f & 1 else d) for r, d, f in zip(hr.tolist(), hd.tolist(), hf.tolist())]
        assert [tuple(map(str, v)) for v in vals] == [tuple(map(str, w)) for w in want]
        k = [i for i, (a, b) in enumerate(zip(hq, ho)) if (a, b) == (1, 0)]
        assert len(k) == 1 and vals[k[0]] == (4080.4507, 4948.0)
        assert np.isnan(hits.ld32.cpu().numpy()[k[0]]).all()


def test_area_repeated_calls_replay_a_graph_and_stay_correct(gpu):
    """Round 4: ld_area keeps a plan per call shape on the panel (buffers + from the second repetition on ONE HIP graph of
    its launches).  Repeated calls with a DEVICE tensor of positions and every SNP a query -- the shape a driver repeats --
    return the popcount scan's hits every time, also when shapes alternate (two thresholds, two flanks), when the first hit
    buffer is too small (the overflow retry re-plans), and a result handed out earlier is not overwritten by a later call."""
    import torch
    from ld_tools_amd import PackedPanel, ld_area, ops, synth

    n, h = 6000, 1008
    p = PackedPanel.from_codes(synth.synth_codes_device(n, h, seed=21))
    pos = torch.as_tensor(synth.synth_positions(n, step=500)).to(p.device)
    shapes = [(20000, "r_square", 0.8), (20000, "r_square", 0.3), (60000, "d_prime", 0.95)]
    want = {}
    old = ops.get_area_path()
    try:
        ops.set_area_path("popcount")
        for sh in shapes:
            want[sh] = ld_area(p, pos, None, *sh, use_graph=False)
    finally:
        ops.set_area_path(old)

    def same(a, b):
        return len(a) == len(b) and torch.equal(a.query, b.query) and torch.equal(a.oppos, b.oppos) and \
            torch.equal(a.ld32.view(torch.int32), b.ld32.view(torch.int32)) and torch.equal(a.offsets, b.offsets)

    kept = []
    for rep in range(4):
        for sh in shapes:
            got = ld_area(p, pos, None, *sh, check_positions=False)
            assert same(got, want[sh]), (rep, sh)
            kept.append((sh, got))
    for sh, got in kept:                                    # earlier results are their own tensors, not views of a plan
        assert same(got, want[sh]), sh
    plans = [v for k, v in p._area_plans.items() if k != "all_rows"]
    assert len(plans) == len(shapes) and all(pl.graph for pl in plans), "every repeated shape should have its graph by now"
    assert want[shapes[1]].band_passes is None and kept[1][1].band_passes > 0
    # a hit buffer that is too small: the retry takes the exact count and the next call replays the larger plan
    small = ld_area(p, pos, None, 20000, "r_square", 0.05, hit_capacity=4096, check_positions=False)
    again = ld_area(p, pos, None, 20000, "r_square", 0.05, hit_capacity=4096, check_positions=False)
    assert len(small) > 4096 and same(small, again)


def test_planes_of_4gib_or_more_use_the_popcount_kernels(gpu):
    """The matrix kernels address the bit plane with 32-bit lane offsets: a plane of 4 GiB or more (3.4 M SNPs x 10 240
    haplotypes here, zero planes) is LDX_E_UNSUPPORTED on an explicit matrix-pipe path and runs the popcount kernel on auto."""
    import torch
    from ld_tools_amd import LdxError, PackedPanel, ld_triangle

    p = PackedPanel.empty(3_400_000, 10240)
    assert p.alt.numel() >= 1 << 32
    for path in ("fp4", "mfma"):
        with pytest.raises(LdxError):
            ld_triangle(p, unit_range=(0, 16), path=path, fmt="k16")
    a = ld_triangle(p, unit_range=(0, 16), fmt="k16")
    b = ld_triangle(p, unit_range=(0, 16), fmt="k16", path="popcount")
    assert torch.equal(a.cells.view(torch.int32), b.cells.view(torch.int32))
    del p
    torch.cuda.empty_cache()


def test_area_band_beyond_the_plan_limits(gpu):
    """A chromosome of more than 4096 tiles (524 288 SNPs): the band plan kernel's tile-start table no longer fits its LDS
    (plain binary search) and the ticket order is not materialised (plain tile order per XCD range); the band's hits still
    equal the popcount scan's."""
    import torch
    from ld_tools_amd import PackedPanel, ld_area, ops, synth

    n, h = 530000, 128
    p = PackedPanel.from_codes(synth.synth_codes_device(n, h, seed=5))
    pos = torch.as_tensor(synth.synth_positions(n, step=500)).to(p.device)
    old = ops.get_area_path()
    try:
        ops.set_area_path("fp4")
        got = ld_area(p, pos, None, 8000, "r_square", 0.6, check_positions=False, use_graph=False)
        ops.set_area_path("popcount")
        want = ld_area(p, pos, None, 8000, "r_square", 0.6, check_positions=False, use_graph=False)
    finally:
        ops.set_area_path(old)
    assert len(got) == len(want) and len(got) > 100000
    assert torch.equal(got.query, want.query) and torch.equal(got.oppos, want.oppos)
    assert torch.equal(got.ld32.view(torch.int32), want.ld32.view(torch.int32)) and torch.equal(got.offsets, want.offsets)
    assert got.band_passes > 4096


def test_area_rejects_unsorted_positions(gpu):
    from ld_tools_amd import LdxError, PackedPanel, ld_area, synth

    p = PackedPanel.from_codes(synth.synth_codes_host(10, 64, seed=1))
    with pytest.raises(LdxError):
        ld_area(p, [5, 4, 6, 7, 8, 9, 10, 11, 12, 13], None, 10)
    with pytest.raises(LdxError):
        ld_area(p, [1, 2, 3], None, 10)


def test_area_paths_agree(gpu):
    """The two ld_area kernels return the same hits, bit for bit, on a panel with clustered and duplicate positions,
    missing codes, monomorphic SNPs, subset and full query lists, flank 0 and a threshold of 0."""
    from ld_tools_amd import PackedPanel, ld_area, ops, synth

    n, h = 2500, 5008
    codes = synth.synth_codes_host(n, h, seed=21, miss=0.001)
    codes[100] = 0                  # monomorphic REF
    codes[101] = 1                  # monomorphic ALT
    codes[102, ::3] = 2             # a third of the haplotypes missing
    rng = np.random.RandomState(4)
    pos = np.cumsum(rng.choice([0, 1, 3, 40, 400, 5000], size=n, p=[0.05, 0.2, 0.2, 0.3, 0.2, 0.05])) + 1
    p = PackedPanel.from_codes(codes)
    cases = [(None, 2000, "r_square", 0.5), (None, 0, "r_square", 0.0), (list(range(0, n, 3)), 30000, "d_prime", 0.95),
             (list(rng.choice(n, 700, replace=False)), 800, "r_square", 0.0), (None, 10 ** 7, "r_square", 0.9)]
    try:
        for queries, flank, measure, thres in cases:
            got = {}
            for path in ("popcount", "mfma", "fp4"):
                ops.set_area_path(path)
                hits = ld_area(p, pos, queries, flank, measure, thres)
                got[path] = (hits.query.cpu().numpy(), hits.oppos.cpu().numpy(), hits.ld32.cpu().numpy().view(np.uint32),
                             hits.n_pairs)
            a = got["popcount"]
            for mm in ("mfma", "fp4"):
                b = got[mm]
                assert a[3] == b[3] and len(a[0]) == len(b[0]), (mm, flank, measure, thres, len(a[0]), len(b[0]))
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), mm
    finally:
        ops.set_area_path("auto")


def test_area_random_cases_both_kernels_and_oracle(gpu):
    """Seeded sweep over ragged shapes (1 SNP, 1 haplotype, sizes either side of the 128-wide tiles), clustered and repeated
    positions, random flanks / thresholds / query subsets: both ld_area kernels against the C oracle's window loop."""
    from ld_tools_amd import PackedPanel, ld_area, ops, synth
    from oracle import c_oracle

    rng = np.random.RandomState(77)
    shapes = [(1, 64), (2, 1), (63, 37), (129, 129), (300, 1008), (513, 2504), (640, 128), (257, 5008)]
    try:
        for case, (n, h) in enumerate(shapes * 2):
            codes = synth.synth_codes_host(n, h, seed=100 + case, miss=0.003 if case % 2 else 0.0)
            pos = np.cumsum(rng.choice([0, 1, 2, 50, 700], size=n, p=[0.1, 0.3, 0.2, 0.3, 0.1])) + 1
            flank = int(rng.choice([0, 1, 5, 120, 3000, 10 ** 6]))
            measure = "r_square" if rng.rand() < 0.5 else "d_prime"
            thres = float(rng.choice([0.0, 0.3, 0.8, 1.0]))
            queries = None if rng.rand() < 0.5 else sorted(rng.choice(n, max(1, n // 3), replace=False).tolist())
            p = PackedPanel.from_codes(codes)
            o = c_oracle.Panel(codes)
            qs = np.arange(n) if queries is None else np.array(queries)
            hq, ho, hr, hd, hf = o.area(pos, qs, flank, 0 if measure == "r_square" else 1, thres, libm_pow=True)
            for path in ("popcount", "mfma", "fp4"):
                ops.set_area_path(path)
                hits = ld_area(p, pos, queries, flank, measure, thres)
                tag = (case, n, h, flank, measure, thres, path)
                assert len(hits) == len(hq), tag
                assert np.array_equal(hits.query.cpu().numpy(), hq) and np.array_equal(hits.oppos.cpu().numpy(), ho), tag
                ld = hits.ld32.cpu().numpy().reshape(-1, 2)
                assert np.array_equal(k_of(ld[:, 0]), np.rint(hr * 1e4).astype(np.int64)), tag
                assert np.array_equal(k_of(ld[:, 1]), np.rint(hd * 1e4).astype(np.int64)), tag
                assert np.array_equal(flags_of(ld), hf), tag
    finally:
        ops.set_area_path("auto")


# ------------------------------------------------------------------ calc_ld drop-in
def test_calc_ld_dropin(gpu, kat):
    from ld_tools_amd.backend.calc_ld import calc_ld, calc_ld_full

    for item in kat["literal"]:
        got = calc_ld(item["g1"], item["g2"])
        assert list(got) == ["r_square", "d_prime", "var_1_alt_freq", "var_2_alt_freq"]
        for k, w in item["expect"].items():
            assert got[k] == w and type(got[k]) is type(w), (item, k, got[k])
        assert str(got) == str(item["expect"])
    from conftest import realise
    for item in kat["tuples"]:                       # all of them, the D' >> 1 ones (values >= 1024) included
        n, n11, a1, r1, a2, r2 = item["counts"]
        g1, g2 = realise(n, n11, a1, r1, a2, r2)
        got, counts, raw, flags = calc_ld_full(g1, g2)
        assert counts == (n, n11, a1, r1, a2, r2)
        assert str(got) == str(item["expect"]), item
        for key, w in item["expect"].items():
            assert got[key] == w and type(got[key]) is type(w), (item, key)
    long = [1, 0] * 20000                            # no LDX_MAX_HAPS limit on the pair-by-pair path
    assert calc_ld(long, long[1:] + [0])["var_1_alt_freq"] == 0.5
    with pytest.raises(ZeroDivisionError):
        calc_ld([], [1])
    assert str(calc_ld(np.array([1, 0, 1, 0]), np.array([1, 1, 0, 0]))) == str(calc_ld([1, 0, 1, 0], [1, 1, 0, 0]))
    # the reference's own import statement (ld_triangle.py:377), repository root on sys.path: the same function
    import importlib
    import sys
    from pathlib import Path
    root = str(Path(__file__).resolve().parent.parent)
    if root not in sys.path:
        sys.path.insert(0, root)
    ref_path = importlib.import_module("backend.calc_ld")
    assert ref_path.calc_ld is calc_ld
    for item in kat["literal"]:
        assert str(ref_path.calc_ld(item["g1"], item["g2"])) == str(item["expect"])


def test_fuzz_matrix_kernels_against_popcount(gpu):
    """tools/gpu_fuzz.py for ten seconds: random panel shapes, missing-code rates, degenerate rows, unit ranges, cell
    formats, repeated launches into poisoned buffers, and random ld_area scans -- the FP4 and int8 matrix-pipe kernels
    against the popcount kernels, bit for bit.  (240 s of it: 4 867 panels, 5.3e10 cells, no difference.)"""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("gpu_fuzz", Path(__file__).resolve().parent.parent / "tools" / "gpu_fuzz.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    summary = mod.run(10.0, seed=20261004)
    assert summary.startswith("fuzz ok")

