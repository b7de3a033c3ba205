"""Deterministic synthetic genotype panels, identical on host and device.

All randomness is a counter-based 64-bit hash; the only floating-point step is the per-SNP ALT
probability, computed once on the host and handed to the device as a 64-bit threshold, so the
numpy generator below and ldx_synth_codes_dev produce the same codes bit for bit.

    key(seed, i, h) = mix64(seed ^ i*0x9E3779B97F4A7C15 ^ h*0xBF58476D1CE4E5B9)   (splitmix64 finaliser)
    p_i             = sin^2(pi/2 * u),  u = key(seed, i, 2^64-1) / 2^64, clamped to [1/H, 1-1/H]
    copy(i, h)      = i % block_len != 0 and key(seed+1, i, h) < rho * 2^64
    g[i][h]         = g[i-1][h] if copy else key(seed+2, i, h) < p_i * 2^64
    code[i][h]      = 2 if key(seed+3, i, h) < miss * 2^64 else g[i][h]

Panels that are not all "ordinary" (round 6; a sub-panel of the ALL-panel variants holds many SNPs that are monomorphic
in it, and the reference maps those to the int 0: calc_ld.py:55-76,89-90):
    miss applies only to rows with key(seed+6, i, 2^64-1) < miss_rows * 2^64   (miss_rows = 1: every row)
    a row with key(seed+4, i, 2^64-1) < mono * 2^64 is monomorphic: every code 0, or every code 1 when
    key(seed+5, i, 2^64-1) & 7 == 0; the LD chain underneath is not disturbed
"""
from __future__ import annotations

from typing import Optional

import numpy as np

M64 = (1 << 64) - 1
_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

BENCH_SEED = 20261003
BLOCK_LEN = 32
RHO = 0.9


def mix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + _G
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        return x ^ (x >> np.uint64(31))


def key64(seed: int, i, h) -> np.ndarray:
    with np.errstate(over="ignore"):
        i = np.asarray(i, dtype=np.uint64)
        h = np.asarray(h, dtype=np.uint64)
        return mix64(np.uint64(seed & M64) ^ (i * _G) ^ (h * _M1))


def prob_to_thr(p: float) -> int:
    """probability -> 64-bit threshold (key < thr  <=>  event), exact for p in [0, 1)."""
    if p <= 0.0:
        return 0
    if p >= 1.0:
        return M64
    return min(M64, int(p * 18446744073709551616.0))


def snp_thresholds(seed: int, first_snp: int, count: int, n_hap: int) -> np.ndarray:
    """uint64 [count]: ALT probability * 2^64 of global SNPs first_snp .. first_snp+count-1."""
    i = np.arange(first_snp, first_snp + count, dtype=np.uint64)
    u = key64(seed, i, np.uint64(M64)).astype(np.float64) / 18446744073709551616.0
    p = np.sin(0.5 * np.pi * u) ** 2
    p = np.clip(p, 1.0 / n_hap, 1.0 - 1.0 / n_hap)
    thr = np.minimum(p * 18446744073709551616.0, float(M64))
    # float -> uint64 conversion of values >= 2^63 is fine in numpy; clip first to stay in range
    return np.minimum(thr, 18446744073709549568.0).astype(np.uint64)


def synth_codes_host(n_snps: int, n_hap: int, seed: int = BENCH_SEED, miss: float = 0.0,
                     block_len: int = BLOCK_LEN, rho: float = RHO, snp_offset: int = 0, mono: float = 0.0,
                     miss_rows: float = 1.0) -> np.ndarray:
    """int8 [n_snps][n_hap] codes of global SNPs [snp_offset, snp_offset + n_snps) (numpy)."""
    first_block = snp_offset // block_len
    last_block = (snp_offset + n_snps - 1) // block_len
    nb = last_block - first_block + 1
    base = first_block * block_len
    thr = snp_thresholds(seed, base, nb * block_len, n_hap).reshape(nb, block_len)
    rho_thr = np.uint64(prob_to_thr(rho))
    miss_thr = np.uint64(prob_to_thr(miss))
    h = np.arange(n_hap, dtype=np.uint64)[None, :]
    g = np.zeros((nb, n_hap), dtype=np.int8)
    out = np.empty((nb * block_len, n_hap), dtype=np.int8)
    for k in range(block_len):
        gi = (base + np.arange(nb, dtype=np.uint64) * np.uint64(block_len) + np.uint64(k))[:, None]
        fresh = (key64(seed + 2, gi, h) < thr[:, k][:, None]).astype(np.int8)
        if k == 0:
            g = fresh
        else:
            copy = key64(seed + 1, gi, h) < rho_thr
            g = np.where(copy, g, fresh)
        code = g
        if miss > 0.0:
            hit = key64(seed + 3, gi, h) < miss_thr
            if miss_rows < 1.0:
                hit = hit & (key64(seed + 6, gi, np.uint64(M64)) < np.uint64(prob_to_thr(miss_rows)))
            code = np.where(hit, np.int8(2), g)
        if mono > 0.0:
            is_mono = key64(seed + 4, gi, np.uint64(M64)) < np.uint64(prob_to_thr(mono))
            all_alt = (key64(seed + 5, gi, np.uint64(M64)) & np.uint64(7)) == np.uint64(0)
            code = np.where(is_mono, np.where(all_alt, np.int8(1), np.int8(0)), code)
        out[k::block_len] = code
    lo = snp_offset - base
    return np.ascontiguousarray(out[lo:lo + n_snps])


def synth_codes_device(n_snps: int, n_hap: int, seed: int = BENCH_SEED, miss: float = 0.0,
                       block_len: int = BLOCK_LEN, rho: float = RHO, snp_offset: int = 0, device=None,
                       mono: float = 0.0, miss_rows: float = 1.0):
    """Same codes generated on the GPU (torch int8 tensor [n_snps][ld], ld = n_hap rounded up to 16)."""
    import torch

    from ._lib import check, lib
    from .panel import _stream_ptr, require_gpu

    dev = device or require_gpu()
    first_block = snp_offset // block_len
    last_block = (snp_offset + n_snps - 1) // block_len
    nb = last_block - first_block + 1
    thr = snp_thresholds(seed, first_block * block_len, nb * block_len, n_hap)
    thr_d = torch.from_numpy(thr.view(np.int64)).to(dev)
    ld = (n_hap + 15) // 16 * 16
    codes = torch.empty((n_snps, ld), dtype=torch.int8, device=dev)
    check(lib.ldx_synth_codes_ex_dev(codes.data_ptr(), n_snps, n_hap, ld, seed & M64, thr_d.data_ptr(),
                                     prob_to_thr(rho), block_len, prob_to_thr(miss), snp_offset, prob_to_thr(mono),
                                     M64 if miss_rows >= 1.0 else prob_to_thr(miss_rows), _stream_ptr()),
          "ldx_synth_codes_ex_dev")
    return codes[:, :n_hap]


def synth_positions(n_snps: int, step: int = 500, first: int = 1) -> np.ndarray:
    """pos_i = first + step*i (100k SNPs over 50 Mb, +-500 kb = +-1000 neighbours)."""
    return first + step * np.arange(n_snps, dtype=np.int64)
