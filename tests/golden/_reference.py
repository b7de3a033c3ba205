"""Load the REAL reference's calc_ld for the golden generators -- by file path, never through sys.path.

This repo has its own top-level ``backend/`` package (the GPU drop-in for ``from backend.calc_ld import
calc_ld``), so an ``import backend.calc_ld`` with /root/reference on sys.path can resolve to the product
instead of the reference.  The generators therefore load ``/root/reference/backend/calc_ld.py`` by path and
check where the function's code object really came from.  Build container only: /root/reference never travels.
"""
from __future__ import annotations

import importlib.util
import sys
import zipfile
from pathlib import Path

import numpy as np

REFERENCE_ROOT = Path("/root/reference")
REFERENCE_CALC_LD = REFERENCE_ROOT / "backend" / "calc_ld.py"


def available() -> bool:
    return REFERENCE_CALC_LD.is_file()


def load_reference_calc_ld():
    """The reference's calc_ld function object (backend/calc_ld.py:3-99)."""
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ldx_reference_calc_ld", str(REFERENCE_CALC_LD))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fn = mod.calc_ld
    where = fn.__code__.co_filename
    assert where.startswith(str(REFERENCE_ROOT) + "/"), f"calc_ld came from {where}, not from the reference"
    assert "ld_tools_amd" not in getattr(fn, "__module__", ""), fn.__module__
    return fn


def save_npz(path, **arrays) -> None:
    """np.savez_compressed with fixed zip timestamps: the same arrays give the same bytes on every run."""
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_DEFLATED, compresslevel=6) as zf:
        for name in sorted(arrays):
            info = zipfile.ZipInfo(name + ".npy", date_time=(1980, 1, 1, 0, 0, 0))
            info.compress_type = zipfile.ZIP_DEFLATED
            info.external_attr = 0o644 << 16
            with zf.open(info, "w", force_zip64=True) as fh:
                np.lib.format.write_array(fh, np.ascontiguousarray(arrays[name]), allow_pickle=False)
