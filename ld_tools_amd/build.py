"""Build libldx.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python ld_tools_amd/build.py [--force] [--save-temps] [--out libldx_x.so] [-DLDX_...]

(run the FILE, not `-m ld_tools_amd.build`: importing the package loads libldx.so first, and a library that is one ABI
behind its header -- the very case a rebuild is for -- then fails the import before the build starts)

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
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libldx.so")
    return exe


def source_digest() -> str:
    """sha256 (16 hex digits) of the kernel sources and headers as they are in this tree: bench.py prints it next to the
    digest recorded with profiles/traffic.json, so a counter record taken from an older kernel is visible as such."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted([CSRC / s for s in SOURCES] + HEADERS):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + HEADERS + [Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, save_temps: bool = False, verbose: bool = True, out: Path = LIB,
          defines=()) -> Path:
    """Build the library.  `out` / `defines` exist for tuning experiments (ablation variants of the
    kernels selected with -DLDX_AB_...; loaded through the LDX_LIB environment variable)."""
    if not force and not stale() and out == LIB:
        return LIB
    tmp_lib = out.with_name(f"{out.name}.tmp.{os.getpid()}")
    cwd = PKG / "build"
    cwd.mkdir(exist_ok=True)
    tag = f"{out.stem}.{os.getpid()}"
    objs = [cwd / f"{Path(s).stem}.{tag}.o" for s in SOURCES]
    compile_flags = [f for f in FLAGS if f != "-shared"]
    extra = ["-save-temps=cwd", "-Rpass-analysis=kernel-resource-usage"] if save_temps else []   # .s / .bc land in build/
    cmds = [[hipcc(), *compile_flags, *defines, *extra, "-c", str(CSRC / s), "-o", str(o)] for s, o in zip(SOURCES, objs)]
    link = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", *[str(o) for o in objs], "-o", str(tmp_lib)]
    if verbose:
        for c in cmds + [link]:
            print("[ldx build]", " ".join(c), file=sys.stderr)
    try:
        # one hipcc per source file, side by side (the matrix kernel's instantiations dominate: minutes in one process)
        from concurrent.futures import ThreadPoolExecutor

        def run(c):
            return subprocess.run(c, cwd=str(cwd), capture_output=not verbose and not save_temps, text=True)

        with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 1)) as pool:
            results = list(pool.map(run, cmds))
        for c, r in zip(cmds, results):
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed ({r.returncode}): {' '.join(c)}\n{r.stderr or ''}")
        subprocess.run(link, check=True, cwd=str(cwd))
        os.replace(tmp_lib, out)   # atomic: a concurrent loader sees the old or the new file, never half
    finally:
        for f in [tmp_lib, *objs]:
            if f.exists():
                f.unlink()
    return out


if __name__ == "__main__":
    out = LIB
    if "--out" in sys.argv:
        out = PKG / sys.argv[sys.argv.index("--out") + 1]
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, out=out,
                defines=[a for a in sys.argv[1:] if a.startswith("-D")]))
