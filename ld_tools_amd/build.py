"""Build libldx.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m ld_tools_amd.build [--force] [--save-temps]

The product never falls back to anything else: if the library is missing and cannot be built,
importing ld_tools_amd._lib raises.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libldx.so"
SOURCES = ["ldx_api.hip", "ldx_pack.hip", "ldx_pairs.hip", "ldx_mfma.hip", "ldx_area.hip", "ldx_synth.hip"]
HEADERS = [CSRC / "ldx_common.h", CSRC / "ldx_tile.h", PKG.parent / "include" / "ldx.h"]

# -ffp-contract=off: the epilogue must round every product and sum separately (calc_ld.py:50);
# hipcc's default for device code is fp-contract=fast.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libldx.so")
    return exe


def stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + HEADERS + [Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, save_temps: bool = False, verbose: bool = True) -> Path:
    if not force and not stale():
        return LIB
    tmp_lib = LIB.with_name(f"libldx.so.tmp.{os.getpid()}")
    cmd = [hipcc(), *FLAGS, *[str(CSRC / s) for s in SOURCES], "-o", str(tmp_lib)]
    if save_temps:
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    if verbose:
        print("[ldx build]", " ".join(cmd), file=sys.stderr)
    try:
        subprocess.run(cmd, check=True, cwd=str(PKG))
        os.replace(tmp_lib, LIB)   # atomic: a concurrent loader sees the old or the new file, never half
    finally:
        if tmp_lib.exists():
            tmp_lib.unlink()
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
    print(LIB)
