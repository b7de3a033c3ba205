"""CPU: the C-ABI library loads and exports what include/ldx.h declares; host-side logic.

No kernel is launched here (there is no GPU in the build container).
"""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def header_functions():
    text = (ROOT / "include" / "ldx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ldx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ld_tools_amd import _lib

    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in ldx.h but not exported by libldx.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in ld_tools_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)
    assert _lib.version() == 102


def test_no_torch_or_python_dependency_in_library():
    """The boundary is plain C: the shared object must not link torch / libpython."""
    import subprocess

    from ld_tools_amd import _lib

    out = subprocess.run(["ldd", str(_lib.LIB_PATH)], capture_output=True, text=True).stdout
    assert "libamdhip64" in out
    assert "torch" not in out and "libpython" not in out


def test_every_workgroup_barrier_drains_lds_first(tmp_path):
    """Every s_barrier in the shipped gfx950 code is preceded by s_waitcnt lgkmcnt(0) with no LDS / scalar-memory
    instruction in between (csrc/ldx_common.h, block_sync).  hipcc leaves that wait out of a plain __syncthreads();
    without it a pass ticket written to LDS just before the barrier was, about once in 10^6 passes, read stale by
    another wave and one pass of a 50 000 x 1008 triangle came back miscomputed (tools/gpu_soak.py)."""
    import shutil
    import subprocess

    from ld_tools_amd import _lib

    objdump = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")
    if not objdump.exists():
        pytest.skip("llvm-objdump not in this image")
    shutil.copy(_lib.LIB_PATH, tmp_path / "libldx.so")            # --offloading unbundles next to its input
    subprocess.run([str(objdump), "--offloading", "libldx.so"], cwd=tmp_path, capture_output=True, check=True)
    objs = sorted(tmp_path.glob("libldx.so.*gfx950"))
    assert len(objs) >= 5, "one gfx950 code object per .hip source"
    stop = re.compile(r"(ds_|s_load|s_buffer_load|flat_|scratch_|s_cbranch|s_branch|s_setpc|s_swappc|s_endpgm)")
    barriers = 0
    for obj in objs:
        text = subprocess.run([str(objdump), "-d", str(obj)], capture_output=True, text=True, check=True).stdout
        lines = [ln.split("//")[0].strip() for ln in text.split("\n")]
        for i, ln in enumerate(lines):
            if not ln.startswith("s_barrier"):
                continue
            barriers += 1
            j = i - 1
            while j >= 0 and not (lines[j].startswith("s_waitcnt") and "lgkmcnt(0)" in lines[j]):
                assert lines[j] and not lines[j].endswith(":") and not stop.match(lines[j]), \
                    f"{obj.name}: s_barrier without a preceding s_waitcnt lgkmcnt(0): {lines[max(0, i - 6): i + 1]}"
                j -= 1
            assert j >= 0
    assert barriers >= 100     # the matrix-pipe kernels alone hold > 100 (one per K-block step, unrolled by three)


def test_no_packed_fp32_broadcasts_in_shipped_code(tmp_path):
    """`v_pk_{mul,fma,add}_f32` that broadcast one register of a VGPR pair (op_sel) returned wrong low halves in lanes
    32-63 about once in 10^4 executions beside another wave's MFMAs on gfx950; they also run no
    faster than the scalar forms there.  The build keeps hipcc from forming them (-fno-slp-vectorize): check."""
    import shutil
    import subprocess

    from ld_tools_amd import _lib

    objdump = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")
    if not objdump.exists():
        pytest.skip("llvm-objdump not in this image")
    shutil.copy(_lib.LIB_PATH, tmp_path / "libldx.so")
    subprocess.run([str(objdump), "--offloading", "libldx.so"], cwd=tmp_path, capture_output=True, check=True)
    for obj in sorted(tmp_path.glob("libldx.so.*gfx950")):
        text = subprocess.run([str(objdump), "-d", str(obj)], capture_output=True, text=True, check=True).stdout
        bad = [ln.split("//")[0].strip() for ln in text.split("\n")
               if re.search(r"v_pk_(mul|fma|add)_f32", ln) and "op_sel" in ln and not re.search(r"\bs\[\d+:\d+\]", ln)]
        assert not bad, f"{obj.name}: {bad[:3]}"


def test_no_vmem_reads_a_valu_written_sgpr_too_early(tmp_path):
    """gfx9 hazard: a VMEM instruction that reads an SGPR written by a VALU instruction (v_readlane_b32 reloading a spilled
    SGPR, v_readfirstlane_b32) needs five wait states in between.  hipcc inserts them for its own instructions, not for
    inline asm: the kernels' hand-issued loads / stores with a scalar base route it through an s_mov_b64 inside the asm
    (csrc/ldx_mfma.hip, gload16_s).  Scan the shipped code object for the pattern (round 4: one instantiation read a
    stale base and faulted)."""
    import shutil
    import subprocess

    from ld_tools_amd import _lib

    objdump = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")
    if not objdump.exists():
        pytest.skip("llvm-objdump not in this image")
    shutil.copy(_lib.LIB_PATH, tmp_path / "libldx.so")
    subprocess.run([str(objdump), "--offloading", "libldx.so"], cwd=tmp_path, capture_output=True, check=True)
    checked = 0
    for obj in sorted(tmp_path.glob("libldx.so.*gfx950")):
        text = subprocess.run([str(objdump), "-d", str(obj)], capture_output=True, text=True, check=True).stdout
        ins = [ln.split("//")[0].strip() for ln in text.split("\n") if ln.startswith("\t")]
        for k, ln in enumerate(ins):
            m = re.match(r"(?:global|buffer|scratch|flat)_\w+ .*?\bs\[(\d+):(\d+)\]", ln)
            if not m:
                continue
            checked += 1
            #  a scalar-base LOAD must not return into its own address register while a second load of the same
            # asm block still reads it: `global_load_dwordx4 v[a:b], vN, s[..]` with a <= N <= b is only legal for the LAST
            # load that uses vN (the asm's outputs are early-clobber, so hipcc never allocates it that way)
            ld = re.match(r"global_load_\w+ v(?:(\d+)|\[(\d+):(\d+)\]), v(\d+), s\[", ln)
            if ld:
                dlo_, dhi_ = int(ld.group(1) or ld.group(2)), int(ld.group(1) or ld.group(3))
                addr = int(ld.group(4))
                nxt = ins[k + 1] if k + 1 < len(ins) else ""
                again = re.match(r"global_load_\w+ v(?:\d+|\[\d+:\d+\]), v(\d+), s\[", nxt)
                assert not (dlo_ <= addr <= dhi_ and again and int(again.group(1)) == addr), \
                    f"{obj.name}: `{ln}` overwrites the address register `{nxt}` still reads"
            lo, hi = int(m.group(1)), int(m.group(2))
            waited = 0
            for prev in reversed(ins[max(0, k - 8):k]):
                if waited >= 5:
                    break
                w = re.match(r"v_(?:readlane|readfirstlane)_b32 s(\d+)", prev)
                assert not (w and lo <= int(w.group(1)) <= hi), f"{obj.name}: `{prev}` {waited} wait states before `{ln}`"
                d = re.match(r"s_\w+ s(?:(\d+)|\[(\d+):(\d+)\])", prev)      # the last writer is a scalar instruction: interlocked
                if d and not prev.startswith(("s_cmp", "s_bitcmp", "s_waitcnt", "s_cbranch", "s_branch", "s_nop", "s_barrier")):
                    dlo = int(d.group(1) or d.group(2))
                    dhi = int(d.group(1) or d.group(3))
                    if dlo <= lo and hi <= dhi:
                        break
                nop = re.match(r"s_nop (\d+)", prev)
                waited += int(nop.group(1)) + 1 if nop else 1
    assert checked > 100   # the scan saw the scalar-base loads and stores


def test_no_scratch_access_inside_the_hand_counted_load_windows(tmp_path):
    """The matrix kernels issue their K-loop loads in inline asm and wait for them with hand-counted `s_waitcnt vmcnt(N)`
    (csrc/ldx_mfma.hip); gfx9 counts EVERY vector-memory operation of a wave in that one counter, so a register spill the
    compiler places inside such a window (a scratch_load / scratch_store it does not know it must not add) shifts the count
    and a fragment is used before it has arrived -- wrong cells, silently.  Round 5 met it twice while editing the kernel
    (an explicit zero C operand: 80 spilled registers; two more scalar kernel arguments: the band kernel).  In every
    instantiation of triangle_mfma_kernel in the shipped code object: no scratch access between the last `vmcnt(0)` in front
    of a K loop's first MFMA and its last MFMA."""
    import shutil
    import subprocess

    from ld_tools_amd import _lib

    objdump = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")
    if not objdump.exists():
        pytest.skip("llvm-objdump not in this image")
    shutil.copy(_lib.LIB_PATH, tmp_path / "libldx.so")
    subprocess.run([str(objdump), "--offloading", "libldx.so"], cwd=tmp_path, capture_output=True, check=True)
    seen = 0
    for obj in sorted(tmp_path.glob("libldx.so.*gfx950")):
        text = subprocess.run([str(objdump), "-d", str(obj)], capture_output=True, text=True, check=True).stdout
        funcs, cur = {}, None
        for ln in text.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
            if m:
                cur = funcs.setdefault(m.group(1), [])
            elif ln.startswith("\t") and cur is not None:
                cur.append(ln.split("//")[0].strip())
        for name, ins in funcs.items():
            if "triangle_mfma_kernel" not in name:
                continue
            mf = [k for k, x in enumerate(ins) if x.startswith("v_mfma")]
            assert mf, name
            clusters = []
            for k in mf:
                if clusters and k - clusters[-1][1] < 300:
                    clusters[-1][1] = k
                else:
                    clusters.append([k, k])
            drains = [k for k, x in enumerate(ins) if re.match(r"s_waitcnt (?:.*\s)?vmcnt\(0\)", x)]
            for a, b in clusters:
                # The window: from the loads at the top of the first K-block (behind the prologue's full drain, at most 100
                # instructions in front of the first MFMA) to the loop's last MFMA.  Behind it only FULL drains follow
                # (`vmcnt(0)` covers whatever is in flight, a reload included), so a reload at the loop's exit is harmless.
                lo = max(max([k for k in drains if k < a], default=0), a - 100)
                hi = b
                bad = [(k, ins[k]) for k in range(lo, hi + 1) if ins[k].startswith("scratch_")]
                assert not bad, f"{name[:70]}: scratch access inside a hand-counted window: {bad[:4]}"
                seen += 1
    assert seen >= 14                                   # every instantiation's K loops were looked at


def test_geometry_helpers():
    from ld_tools_amd import _lib, dist

    L = _lib.lib
    for n in (1, 2, 127, 128, 129, 1000, 10000, 100000):
        T = (n + 127) // 128
        assert L.ldx_n_slabs(n) == T and L.ldx_padded_snps(n) == T * 128
        assert L.ldx_triangle_units(n) == dist.triangle_units(n)
        # every valid pair maps into the unit range, and tile bases are increasing
        assert L.ldx_triangle_tile_base(n, 0) == 0
        assert L.ldx_triangle_tile_base(n, T) == L.ldx_triangle_units(n)
        if n > 1:
            assert L.ldx_triangle_unit_of(n, n - 1, 0) == (n - 1) // 8
            assert L.ldx_triangle_unit_of(n, n - 1, n - 2) < L.ldx_triangle_units(n)
    for h in (1, 128, 129, 256, 257, 1008, 5008, 10240):     # 128-haplotype chunks, allocated in pairs
        assert L.ldx_n_chunks(h) == 2 * ((h + 255) // 256) == dist.n_chunks(h)
    assert L.ldx_plane_bytes(10000, 5008) == 79 * 40 * 128 * 16
    # cells >= pairs, with less than 4 % padding at the bench size
    cells = L.ldx_triangle_units(10000) * 1024
    assert 10000 * 9999 // 2 <= cells < 1.04 * 10000 * 9999 // 2
    # round 6: the matrix kernel's pass scheduler lives in a caller-owned workspace of a fixed, 256-byte-granular size
    assert L.ldx_triangle_workspace_bytes() % 256 == 0 and 256 <= L.ldx_triangle_workspace_bytes() <= 1 << 16
    # ... and a rank's share of the work in PAIRS (bench.py's per_rank record): the unit ranges tile the triangle exactly
    for n, world in ((300, 3), (1000, 4), (10000, 8)):
        parts = dist.unit_partition(n, world)
        pairs = [dist.pairs_in_units(n, a, b) for a, b in parts]
        assert sum(pairs) == n * (n - 1) // 2 and max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
        rows, _ = dist.unit_cells(n, *parts[1])
        assert len(rows) == pairs[1]


def test_unit_cell_order_is_a_permutation_and_matches_the_library():
    """LDX_CELL_OFFSET (include/ldx.h): every (row % 8, column % 128) gets its own element of the unit's 1024; the library's
    ldx_triangle_cell_index and the Python helper agree (every producer and consumer of strip output goes through them)."""
    from ld_tools_amd import _lib

    r, c = np.meshgrid(np.arange(8), np.arange(128), indexing="ij")
    L = _lib.lib
    for fmt in ("k16", "ld32", "k16r", "k16d"):
        off = _lib.cell_offset(r, c, fmt)
        assert sorted(off.ravel().tolist()) == list(range(1024))
        if fmt in _lib.ONE_MEASURE:        # the one-measure formats (round 6) order their 2-byte cells like the 4-byte format
            assert np.array_equal(off, _lib.cell_offset(r, c, "k16")) and _lib.ONE_MEASURE_FMT[_lib.ONE_MEASURE[fmt]] == fmt
            for n, i, j in ((10000, 9999, 3), (300, 299, 298), (100000, 77777, 12345)):
                assert L.ldx_triangle_cell_index(n, i, j, _lib.FORMATS[fmt]) == L.ldx_triangle_cell_index(n, i, j, _lib.FORMATS["k16"])
            continue
        # the four columns one lane of the matrix kernel holds (l, l + 32, l + 64, l + 96) are adjacent: all four (4-byte cells)
        # or in two pairs (8-byte cells), so that the lane writes 16 bytes per store
        for row in range(8):
            for lane in range(32):
                q = [int(_lib.cell_offset(row, lane + 32 * tt, fmt)) for tt in range(4)]
                if fmt == "k16":
                    assert q == [q[0] + k for k in range(4)] and q[0] % 4 == 0
                else:
                    assert q[1] == q[0] + 1 and q[3] == q[2] + 1 and q[0] % 2 == 0 and q[2] == q[0] + 64
        for n in (130, 1000, 10000):
            for row, col in ((1, 0), (n - 1, 0), (n - 1, n - 2), (129, 127), (n // 2, n // 3)):
                u = L.ldx_triangle_unit_of(n, row, col)
                assert L.ldx_triangle_cell_index(n, row, col, _lib.FORMATS[fmt]) == \
                    u * 1024 + int(_lib.cell_offset(row % 8, col % 128, fmt))


def test_device_calls_fail_loudly_without_gpu():
    """No GPU here: the product must raise, not compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ld_tools_amd import LdxError, PackedPanel
    from ld_tools_amd.backend.calc_ld import calc_ld

    with pytest.raises(LdxError):
        PackedPanel.from_codes(np.zeros((4, 8), dtype=np.int8))
    with pytest.raises(LdxError):
        calc_ld([1, 0, 1, 0], [1, 1, 0, 0])
    with pytest.raises(ZeroDivisionError):       # calc_ld.py:33 behaviour is decided before any device call
        calc_ld([], [])


def test_reference_import_lines_work_unchanged():
    """the drop-in keeps the reference's MODULE PATH.  The three scripts import the hot path with
    `from backend.calc_ld import calc_ld` (ld_triangle.py:377, ld_area.py:309, ld_lite.py:61) and the two lookups with
    `from backend.get_sample_names import get_sample_names` / `from backend.create_src_dict import create_src_dict`
    (ld_triangle.py:373-374): with the repository root on sys.path those statements run as they stand and bind the GPU
    implementation (no CPU code behind them: the call raises without a GPU)."""
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from backend.calc_ld import calc_ld\n"                               # ld_triangle.py:377, verbatim
        "from backend.get_sample_names import get_sample_names\n"             # ld_triangle.py:373
        "from backend.create_src_dict import create_src_dict\n"               # ld_triangle.py:374
        "from backend.prep_intgen_data import prep_intgen_data\n"             # ld_triangle.py:372 (returns the prepared folder's db)
        "import tempfile, os\n"
        "d = tempfile.mkdtemp(); open(os.path.join(d, 'conversion.db'), 'w').close()\n"
        "assert prep_intgen_data(d) == os.path.join(d, 'conversion.db')\n"
        "try:\n"
        "    prep_intgen_data(tempfile.mkdtemp())\n"
        "    raise SystemExit('an unprepared folder must raise')\n"
        "except FileNotFoundError:\n"
        "    pass\n"
        "import ld_tools_amd.backend.calc_ld as impl, inspect\n"
        "assert calc_ld is impl.calc_ld\n"
        "assert list(inspect.signature(calc_ld).parameters) == ['var_1_genotypes', 'var_2_genotypes']\n"
        "import torch\n"
        "if not torch.cuda.is_available():\n"
        "    from ld_tools_amd import LdxError\n"
        "    try:\n"
        "        calc_ld([1, 0, 1, 0], [1, 1, 0, 0])\n"
        "    except LdxError:\n"
        "        print('raises')\n"
        "else:\n"
        "    print(calc_ld([1, 0, 1, 0, 1, 1, 0, 0], [1, 0, 1, 0, 0, 1, 0, 1]))\n"
    ) % str(ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/", timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() in ("raises", "{'r_square': 0.25, 'd_prime': 0.5, 'var_1_alt_freq': 0.5, 'var_2_alt_freq': 0.5}")


def test_encode_codes_matches_list_count_semantics():
    from ld_tools_amd import encode_codes

    seq = [1, 0, None, 2, 1.0, 0.0, True, False, "1", float("nan"), -1, 1.5]
    got = encode_codes(seq).tolist()
    want = [1 if v == 1 else (0 if v == 0 else 2) for v in seq]
    assert got == want == [1, 0, 2, 2, 1, 0, 1, 0, 2, 2, 2, 2]
    assert encode_codes(seq).tolist().count(1) == seq.count(1)
    assert encode_codes(seq).tolist().count(0) == seq.count(0)
    # vectorised paths
    assert encode_codes(np.array([0, 1, 2, 3, 1], dtype=np.int64)).tolist() == [0, 1, 2, 2, 1]
    assert encode_codes(np.array([[0.0, 1.0], [np.nan, 2.0]])).tolist() == [[0, 1], [2, 2]]
    assert encode_codes((1, 0, 1)).tolist() == [1, 0, 1]
    # every list shape takes its own path (bytes() for small ints, numeric arrays, objects) and all follow list.count
    for seq in ([1, 0] * 50, ["1", 1, 0], [300, 1, 0, -1], [2.0, 1.0, 0.5], (1, 0, "x"), [True, 1, False, 255, 256]):
        assert encode_codes(seq).tolist() == [1 if v == 1 else (0 if v == 0 else 2) for v in seq], seq
    assert encode_codes([[1, None], [0, 1]]).tolist() == [[1, 2], [0, 1]]
    assert encode_codes([]).tolist() == [] and encode_codes([]).dtype == np.int8


def test_f32_cell_value_from_k_is_the_nearest_float32():
    """ldx_common.h f32_k_to_value: q = k * fl(1e-4); r = fma(-q, 1e4, k); q' = fma(r, fl(1e-4), q) must be the float32
    nearest to k / 10^4 for every k the 4-byte cell can hold.  Emulated with exact rationals (one rounding per operation)."""
    from fractions import Fraction

    def rn32(x: Fraction) -> Fraction:          # round an exact rational to float32 (the values here are far from overflow)
        if x == 0:
            return Fraction(0)
        f = np.float32(float(x))                 # float(x) is correctly rounded to double; a second rounding to float32 can
        lo, hi = np.nextafter(f, np.float32(-np.inf)), np.nextafter(f, np.float32(np.inf))   # differ from direct rounding
        best = min((lo, f, hi), key=lambda c: (abs(Fraction(float(c)) - x), int(np.float32(c).view(np.uint32)) & 1))
        return Fraction(float(best))

    c4 = Fraction(float(np.float32(1e-4)))
    for k in list(range(0, 12000)) + list(range(12000, 32768, 7)) + [32766, 32767]:
        kk = Fraction(k)
        q = rn32(kk * c4)
        r = rn32(-q * 10000 + kk)
        got = rn32(r * c4 + q)
        want = Fraction(float(np.float32(k / 10000.0)))     # k / 1e4 in double, then float32: the encoders' definition
        assert got == want, k


def test_unit_partition_covers_every_pair_once():
    from ld_tools_amd import dist

    for n, world in [(300, 1), (300, 2), (1000, 8), (129, 3), (5, 4)]:
        seen = np.zeros((n, n), dtype=np.int32)
        parts = dist.unit_partition(n, world)
        assert parts[0][0] == 0 and parts[-1][1] == dist.triangle_units(n)
        for (u0, u1) in parts:
            r, c = dist.unit_cells(n, u0, u1)
            np.add.at(seen, (r, c), 1)
        want = np.tril(np.ones((n, n), dtype=np.int32), -1)
        assert np.array_equal(seen, want)
        sizes = [u1 - u0 for (u0, u1) in parts]
        assert max(sizes) - min(sizes) <= 1


def test_slab_partition():
    from ld_tools_amd import dist

    for n, world in [(100000, 8), (10000, 8), (300, 2), (129, 4), (64, 2)]:
        parts = dist.slab_partition(n, world)
        assert parts[0][0] == 0 and parts[-1][1] == n
        for (b0, e0), (b1, e1) in zip(parts, parts[1:]):
            assert e0 == b1
        for b, e in parts[:-1]:
            assert (b % 128 == 0 or b == n) and (e % 128 == 0 or e == n)
        slabs = [(e - b + 127) // 128 for b, e in parts]
        big = max(slabs)                      # every rank holds `big` slabs until they run out (fused_gather relies on it)
        k = next((i for i, x in enumerate(slabs) if x < big), world)
        assert all(x == big for x in slabs[:k]) and all(x == 0 for x in slabs[k + 1:])


def test_query_partition_tiles_the_query_list_and_balances_pairs():
    from ld_tools_amd import dist

    rng = np.random.RandomState(9)
    pos = np.cumsum(rng.choice([0, 1, 10, 900], size=5000, p=[0.05, 0.45, 0.4, 0.1])) + 1
    for queries, flank, world in [(None, 5000, 8), (None, 0, 3), (list(range(0, 5000, 7)), 20000, 4), ([4, 2, 4999], 50, 8),
                                  ([], 50, 2), (None, 10 ** 9, 5)]:
        parts = dist.query_partition(pos, queries, flank, world)
        nq = 5000 if queries is None else len(queries)
        assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == nq
        assert all(e0 == b1 for (_, e0), (b1, _) in zip(parts, parts[1:])) and all(b <= e for b, e in parts)
        if nq >= 100 * world:
            q = np.arange(5000) if queries is None else np.sort(np.array(queries))
            lo = np.searchsorted(pos, np.maximum(pos[q] - flank, 0), side="right")
            hi = np.searchsorted(pos, pos[q] + flank, side="right")
            cost = (hi - lo) + 1
            share = [int(cost[b:e].sum()) for b, e in parts]
            assert max(share) <= 1.1 * sum(share) / world + int(cost.max())


def test_synth_thresholds_and_positions():
    from ld_tools_amd import synth

    thr = synth.snp_thresholds(7, 0, 1000, 5008)
    assert thr.dtype == np.uint64 and thr.min() > 0
    p = thr.astype(np.float64) / 2.0 ** 64
    assert p.min() >= 1 / 5008 - 1e-12 and p.max() <= 1 - 1 / 5008 + 1e-12
    assert 0.35 < p.mean() < 0.65          # U-shaped around one half
    pos = synth.synth_positions(5)
    assert pos.tolist() == [1, 501, 1001, 1501, 2001]
    assert synth.prob_to_thr(0.0) == 0 and synth.prob_to_thr(1.0) == 2 ** 64 - 1


def test_bound_behind_the_few_missing_codes_class():
    """csrc/ldx_common.h, snp_class (round 6): the fast epilogue tiers take SNPs with m = n - a - r <= r / 8 missing codes as
    ordinary because, exactly (rational arithmetic), D' and r^2 never exceed (1 + m1 / r1)(1 + m2 / r2) -- with m <= r / 8:
    1.266, so 10^4 value fits the 15-bit cell and every guard the kClean variants drop.  Every feasible joint table for n <= 13
    (the 3 x 3 table of two SNPs with codes ALT / REF / neither exists iff max(0, a1 + a2 - n) <= n11 <= min(a1, a2)), in both
    sign branches of calc_ld.py:63-76; the bound is attained."""
    from fractions import Fraction as F

    worst, count = F(0), 0
    for n in range(2, 14):
        for a1 in range(1, n):
            for r1 in range(1, n - a1 + 1):
                for a2 in range(1, n):
                    for r2 in range(1, n - a2 + 1):
                        bound = (1 + F(n - a1 - r1, r1)) * (1 + F(n - a2 - r2, r2))
                        for n11 in range(max(0, a1 + a2 - n), min(a1, a2) + 1):
                            dn = n * n11 - a1 * a2
                            if dn == 0:
                                continue
                            b = min(a1 * r2, r1 * a2) if dn > 0 else min(a1 * a2, r1 * r2)      # calc_ld.py:64-65 / :71-72 in counts
                            dp, rsq = F(abs(dn), b), F(dn * dn, a1 * r1 * a2 * r2)                 # :66 / :73, :87-88
                            assert dp <= bound and rsq <= bound, (n, a1, r1, a2, r2, n11)
                            worst = max(worst, dp / bound, rsq / bound)
                            count += 1
    assert worst == 1 and count > 50000
    assert (1 + F(1, 8)) ** 2 < F(127, 100)
