"""ctypes binding of libldx.so (include/ldx.h).  No fallback: a missing library is an error.

The library is built in-tree by ``python ld_tools_amd/build.py`` (``__graft_entry__.build()``
does it).  If it is missing and hipcc is present it is built on first import; if that fails,
the import raises -- there is no CPU path in this package.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from . import build as _build

LIB_PATH = Path(os.environ.get("LDX_LIB") or Path(__file__).resolve().parent / "libldx.so")   # LDX_LIB: tuning variants

# constants of include/ldx.h
SLAB_ROWS = 128
GROUP_ROWS = 8
UNIT_PAIRS = SLAB_ROWS * GROUP_ROWS


def cell_offset(r8, c, fmt="k16"):
    """LDX_CELL_OFFSET of include/ldx.h: element of cell (row % 8, column % 128) inside its unit, in the order of the cell
    format: rows one after the other; inside a row the columns {c0, c0 + 32, c0 + 64, c0 + 96} one lane of the matrix kernel
    holds are adjacent (k16: all four; ld32: in two pairs).  Works on ints and numpy arrays."""
    if fmt in ("k16", "k16r", "k16d"):
        return r8 * SLAB_ROWS + ((c & 31) << 2) + (c >> 5)
    if fmt != "ld32":
        raise ValueError(f"unknown cell format {fmt!r}")
    return r8 * SLAB_ROWS + ((c >> 6) << 6) + ((c & 31) << 1) + ((c >> 5) & 1)


MAX_HAPS = 10240
FLAG_DPRIME_INT0 = 1
FLAG_RSQ_INT0 = 2
MEASURES = {"r_square": 0, "d_prime": 1}
FORMATS = {"ld32": 0, "k16": 1, "k16r": 2, "k16d": 3}   # LDX_OUT_LD32 / LDX_OUT_K16 / LDX_OUT_K16_RSQ / LDX_OUT_K16_DPRIME
ONE_MEASURE = {"k16r": "r_square", "k16d": "d_prime"}    # the one-measure formats (2-byte cells) and what they hold
ONE_MEASURE_FMT = {v: k for k, v in ONE_MEASURE.items()}
LD32_BIG_BITS = 0x7FC00B16               # ldx_ld32 escape (value >= 1024): a quiet NaN
K16_INT0 = 0x8000
K16_BIG = 0x7FFF
INVALID_ROW = 0xFFFFFFFF

E_NAMES = {-1: "LDX_E_ARG", -2: "LDX_E_HIP", -3: "LDX_E_UNSUPPORTED", -4: "LDX_E_OVERFLOW"}


class LdxError(RuntimeError):
    pass


def _load() -> C.CDLL:
    # an existing library is loaded as it is (ranks of one job must not rebuild it under each other);
    # `python ld_tools_amd/build.py` / __graft_entry__.build() refresh a stale one explicitly
    if not LIB_PATH.exists():
        try:
            _build.build(verbose=False)
        except Exception as exc:  # noqa: BLE001
            if not LIB_PATH.exists():
                raise ImportError(
                    f"ld_tools_amd: {LIB_PATH} is missing and could not be built ({exc}); "
                    "run `python ld_tools_amd/build.py` on a machine with hipcc. "
                    "There is no CPU fallback.") from exc
    _share_hip_runtime()
    return C.CDLL(str(LIB_PATH))


def _share_hip_runtime() -> None:
    """Make libldx.so bind to the HIP runtime torch uses (one runtime per process).

    torch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64; a second copy from /opt/rocm in the
    same process cannot open the device ("no ROCm-capable device is detected").  libldx.so only NEEDs
    the soname libamdhip64.so.7, so loading torch's copy first (RTLD_GLOBAL) makes the dynamic loader
    resolve it to that one.  Without torch (a plain C host) the library's RUNPATH finds /opt/rocm.
    """
    try:
        import torch  # noqa: F401
    except ImportError:
        return
    cand = Path(torch.__file__).resolve().parent / "lib" / "libamdhip64.so"
    if cand.exists():
        C.CDLL(str(cand), mode=C.RTLD_GLOBAL)


lib = _load()

_vp, _u32, _u64, _i64, _sz, _int, _dbl = (C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64, C.c_size_t,
                                          C.c_int, C.c_double)

# name -> (restype, argtypes); also the list the symbol test walks (kept in sync with ldx.h)
SIGNATURES = {
    "ldx_version": (_int, []),
    "ldx_last_error": (C.c_char_p, []),
    "ldx_device_count": (_int, []),
    "ldx_device_arch": (_int, [_int, C.c_char_p, _sz]),
    "ldx_n_slabs": (_u32, [_u32]),
    "ldx_n_chunks": (_u32, [_u32]),
    "ldx_plane_bytes": (_sz, [_u32, _u32]),
    "ldx_padded_snps": (_u32, [_u32]),
    "ldx_triangle_units": (_u64, [_u32]),
    "ldx_triangle_unit_of": (_u64, [_u32, _u32, _u32]),
    "ldx_triangle_tile_base": (_u64, [_u32, _u32]),
    "ldx_triangle_cell_index": (_u64, [_u32, _u32, _u32, C.c_int]),
    "ldx_pack_codes_dev": (_int, [_vp, _u32, _u32, _sz, _vp, _vp, _vp, _vp, _vp]),
    "ldx_tile_plane_dev": (_int, [_vp, _u32, _u32, _sz, _vp, _vp, _vp]),
    "ldx_snp_stats_dev": (_int, [_vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "ldx_alt_freq4_dev": (_int, [_vp, _u32, _u32, _vp, _vp]),
    "ldx_pair_counts_dev": (_int, [_vp, _u32, _vp, _u32, _u32, _vp, _sz, _vp]),
    "ldx_ld_from_counts_dev": (_int, [_u32, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ldx_triangle_dev": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u64, _u64, _vp, _vp, _vp, _vp]),
    "ldx_triangle_ex_dev": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u64, _u64, _int, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ldx_triangle_workspace_bytes": (_sz, []),
    "ldx_triangle_workspace_init_dev": (_int, [_vp, _sz, _vp]),
    "ldx_triangle_dense_ex_dev": (_int, [_vp, _int, _u32, _int, _int, _dbl, _u32, _u32, _vp, _sz, _vp]),
    "ldx_ld_from_counts_ex_dev": (_int, [_u32, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ldx_ld_pairs_dev": (_int, [_vp, _vp, _vp, _u32, _u32, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "ldx_debug_force_short_passes": (_int, [_int]),
    "ldx_debug_counters": (_int, [_vp, _int]),
    "ldx_set_triangle_path": (_int, [_int]),
    "ldx_get_triangle_path": (_int, []),
    "ldx_triangle_dense_dev": (_int, [_vp, _u32, _int, _int, _dbl, _u32, _u32, _vp, _sz, _vp]),
    "ldx_area_dev": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _u32, _i64, _int, _dbl, _vp, _u64,
                            _vp, _vp, _sz, _vp]),
    "ldx_area_scan_dev": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _u32, _i64, _int, _dbl, _vp, _u64,
                                 _vp, _vp, _vp, _sz, _vp]),
    "ldx_area_workspace_bytes": (_sz, [_u32, _u32, _u32]),
    "ldx_area_finish_dev": (_int, [_vp, _vp, _u64, _u32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ldx_area_finish_ex_dev": (_int, [_vp, _vp, _u64, _u32, _vp, _vp, _vp, _vp, _sz, _int, _vp]),
    "ldx_area_finish_counts": (_vp, [_vp]),
    "ldx_area_results_dev": (_int, [_vp, _u64, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    "ldx_area_finish_workspace_bytes": (_sz, [_u32]),
    "ldx_area_band_passes_offset": (_sz, [_u32]),
    "ldx_set_area_path": (_int, [_int]),
    "ldx_get_area_path": (_int, []),
    "ldx_synth_codes_dev": (_int, [_vp, _u32, _u32, _sz, _u64, _vp, _u64, _u32, _u64, _u32, _vp]),
    "ldx_synth_codes_ex_dev": (_int, [_vp, _u32, _u32, _sz, _u64, _vp, _u64, _u32, _u64, _u32, _u64, _u64, _vp]),
    "ldx_calc_ld_host": (_int, [_vp, _u32, _vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "ldx_probe_andpop_dev": (_int, [_vp, _u32, _u32, _u32, _vp]),
    "ldx_probe_mfma_dev": (_int, [_vp, _u32, _u32, _u32, _int, _vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here == header and library out of sync
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc: int, what: str = "") -> None:
    """Raise LdxError for a negative return code of an ldx_* call."""
    if rc < 0:
        msg = lib.ldx_last_error().decode("utf-8", "replace")
        raise LdxError(f"{what or 'ldx call'} failed: {E_NAMES.get(rc, rc)}: {msg}")


def version() -> int:
    return lib.ldx_version()
