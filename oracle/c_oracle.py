"""ctypes loader for the C oracle (oracle/ld_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Builds oracle/_build/libldoracle.so with gcc on first use (``make -C oracle``).  Imported only
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "_build" / "libldoracle.so"


def build(force: bool = False) -> Path:
    src = HERE / "ld_oracle.c"
    if force or not LIB.exists() or LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "-s"] + (["-B"] if force else []), check=True)
    return LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
        _lib.ldo_round4.restype = C.c_double
        _lib.ldo_round4.argtypes = [C.c_double]
        _lib.ldo_ld_from_counts.restype = C.c_uint
        _lib.ldo_ld_from_counts.argtypes = [C.c_uint32] * 6 + [C.c_int, C.POINTER(C.c_double)]
        _lib.ldo_area.restype = C.c_size_t
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def round4(x: float) -> float:
    return lib().ldo_round4(float(x))


def ld_from_counts(n, n11, a1, r1, a2, r2, libm_pow=True):
    """-> (rsq_raw, dp_raw, rsq_rnd, dp_rnd, fa1, fa2, flags)"""
    out = (C.c_double * 6)()
    f = lib().ldo_ld_from_counts(n, n11, a1, r1, a2, r2, int(libm_pow), out)
    return (*[out[k] for k in range(6)], f)


def ld_from_counts_v(n, n11, a1, r1, a2, r2, libm_pow=True):
    arrs = [np.ascontiguousarray(x, dtype=np.uint32) for x in (n11, a1, r1, a2, r2)]
    m = arrs[0].size
    rsq_raw, dp_raw, rsq_rnd, dp_rnd = (np.empty(m, dtype=np.float64) for _ in range(4))
    flags = np.empty(m, dtype=np.uint8)
    lib().ldo_ld_from_counts_v(C.c_uint32(n), C.c_size_t(m), *[_p(a) for a in arrs], C.c_int(int(libm_pow)),
                               _p(rsq_raw), _p(dp_raw), _p(rsq_rnd), _p(dp_rnd), _p(flags))
    return rsq_raw, dp_raw, rsq_rnd, dp_rnd, flags


class Panel:
    """Row-major packed panel on the host (uint64 planes) with per-SNP counts."""

    def __init__(self, codes: np.ndarray):
        codes = np.ascontiguousarray(codes, dtype=np.int8)
        self.n_snps, self.n_hap = codes.shape
        self.w64 = (self.n_hap + 63) // 64
        self.alt = np.empty((self.n_snps, self.w64), dtype=np.uint64)
        self.ref = np.empty((self.n_snps, self.w64), dtype=np.uint64)
        L = lib()
        L.ldo_pack(_p(codes), C.c_size_t(self.n_snps), C.c_size_t(self.n_hap), _p(self.alt), _p(self.ref),
                   C.c_size_t(self.w64))
        self.acnt = np.empty(self.n_snps, dtype=np.uint32)
        self.rcnt = np.empty(self.n_snps, dtype=np.uint32)
        L.ldo_counts(_p(self.alt), C.c_size_t(self.n_snps), C.c_size_t(self.w64), _p(self.acnt))
        L.ldo_counts(_p(self.ref), C.c_size_t(self.n_snps), C.c_size_t(self.w64), _p(self.rcnt))

    def pair_counts(self, r0, r1, c0, c1) -> np.ndarray:
        out = np.empty((r1 - r0, c1 - c0), dtype=np.uint32)
        lib().ldo_pair_counts(_p(self.alt), C.c_size_t(self.w64), C.c_size_t(r0), C.c_size_t(r1), C.c_size_t(c0),
                              C.c_size_t(c1), _p(out))
        return out

    def triangle(self, row0=0, row1=None, libm_pow=True, want=("n11", "rsq_raw", "dp_raw", "rsq_rnd", "dp_rnd",
                                                                "flags")):
        """Dense [n][n] outputs (strict lower triangle written, rest zero) for rows [row0, row1)."""
        n = self.n_snps
        row1 = n if row1 is None else row1
        bufs = {
            "n11": np.zeros((n, n), dtype=np.uint32) if "n11" in want else None,
            "rsq_raw": np.zeros((n, n), dtype=np.float64) if "rsq_raw" in want else None,
            "dp_raw": np.zeros((n, n), dtype=np.float64) if "dp_raw" in want else None,
            "rsq_rnd": np.zeros((n, n), dtype=np.float64) if "rsq_rnd" in want else None,
            "dp_rnd": np.zeros((n, n), dtype=np.float64) if "dp_rnd" in want else None,
            "flags": np.zeros((n, n), dtype=np.uint8) if "flags" in want else None,
        }
        lib().ldo_triangle(_p(self.alt), _p(self.acnt), _p(self.rcnt), C.c_size_t(n), C.c_size_t(self.w64),
                           C.c_uint32(self.n_hap), C.c_size_t(row0), C.c_size_t(row1), C.c_int(int(libm_pow)),
                           _p(bufs["n11"]), _p(bufs["rsq_raw"]), _p(bufs["dp_raw"]), _p(bufs["rsq_rnd"]),
                           _p(bufs["dp_rnd"]), _p(bufs["flags"]))
        return bufs

    def triangle_band(self, row0, row1, libm_pow=True, want=("n11", "rsq_rnd", "dp_rnd", "flags")):
        """The same loop for rows [row0, row1) into [row1 - row0][n] arrays (element [i - row0][j], j < i; rest zero):
        what a test needs to walk a 50 000-SNP triangle band by band from several threads without [n][n] buffers.
        ldo_triangle addresses cell (i, j) as base[i * n + j]; the base handed over is the band's first element minus
        row0 * n elements, so only the band's rows are ever touched."""
        n = self.n_snps
        kinds = {"n11": np.uint32, "rsq_raw": np.float64, "dp_raw": np.float64, "rsq_rnd": np.float64,
                 "dp_rnd": np.float64, "flags": np.uint8}
        bufs, ptrs = {}, []
        for name, dt in kinds.items():
            if name in want:
                a = np.zeros((row1 - row0, n), dtype=dt)
                bufs[name] = a
                ptrs.append(C.c_void_p(a.ctypes.data - row0 * n * a.itemsize))
            else:
                bufs[name] = None
                ptrs.append(None)
        lib().ldo_triangle(_p(self.alt), _p(self.acnt), _p(self.rcnt), C.c_size_t(n), C.c_size_t(self.w64),
                           C.c_uint32(self.n_hap), C.c_size_t(row0), C.c_size_t(row1), C.c_int(int(libm_pow)), *ptrs)
        return bufs

    def area(self, positions, queries, flank, measure=0, thres=0.8, libm_pow=True, cap=None):
        positions = np.ascontiguousarray(positions, dtype=np.int64)
        queries = np.ascontiguousarray(queries, dtype=np.uint32)
        cap = cap or max(1024, 4096 * len(queries))
        hq = np.empty(cap, dtype=np.uint32)
        ho = np.empty(cap, dtype=np.uint32)
        hr = np.empty(cap, dtype=np.float64)
        hd = np.empty(cap, dtype=np.float64)
        hf = np.empty(cap, dtype=np.uint8)
        n = lib().ldo_area(_p(self.alt), _p(self.acnt), _p(self.rcnt), C.c_size_t(self.n_snps), C.c_size_t(self.w64),
                           C.c_uint32(self.n_hap), _p(positions), _p(queries), C.c_size_t(len(queries)),
                           C.c_int64(flank), C.c_int(measure), C.c_double(thres), C.c_int(int(libm_pow)),
                           C.c_size_t(cap), _p(hq), _p(ho), _p(hr), _p(hd), _p(hf))
        if n > cap:
            return self.area(positions, queries, flank, measure, thres, libm_pow, cap=int(n))
        return hq[:n], ho[:n], hr[:n], hd[:n], hf[:n]
