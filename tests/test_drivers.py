"""Drivers either side of the hot path: ingest, ld_triangle / ld_area / ld_lite drivers, writers, CLI.

Expected texts: tests/golden/driver_text.json, made by tests/golden/make_golden_drivers.py with the reference's own
calc_ld inside the reference's loop order (tests/ref_loops.py).  CPU tests pin the restated loops (with the oracle's
calc_ld) to that golden and cover the host-only pieces; GPU tests compare what the batched drivers write, byte for
byte.
"""
import json
import os
import sqlite3
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))

import fakevcf  # noqa: E402
import ref_loops  # noqa: E402
from oracle import ld_oracle as orc  # noqa: E402

GOLD = json.loads((ROOT / "tests" / "golden" / "driver_text.json").read_text())


def _rows(vcf):
    rows = [[r.pos, r.id] for r in vcf.records if r.id.startswith("rs") and ";" not in r.id]
    seen, uniq = set(), []
    for r in rows:
        if r[1] not in seen:
            seen.add(r[1])
            uniq.append(r)
    return uniq


@pytest.fixture(scope="module")
def chrom6():
    vcf, names = fakevcf.make_chromosome()
    uniq = _rows(vcf)
    return vcf, names, uniq[::2][:16], uniq[::3]


AREA_CASES = [("r_square", 0.8, 1000), ("d_prime", 0.9, 400), ("r_square", 0.05, 2500)]


# ------------------------------------------------------------------------------------------ CPU
def test_restated_loops_with_oracle_match_reference_golden(chrom6):
    vcf, names, tri_rows, queries = chrom6
    for key, text in GOLD["triangle"].items():
        measure, thres = key.split("|")
        thres = None if thres == "None" else float(thres)
        assert ref_loops.triangle_tsv(vcf, "6", tri_rows, names, measure, thres, ("EUR", "AMR"), ("male", "female"),
                                      orc.calc_ld_lists) == text
    for measure, thres, flank in AREA_CASES:
        got = ref_loops.area_files(vcf, "6", queries, names, flank, measure, thres, "tsv", ("ALL",), ("female",),
                                   orc.calc_ld_lists)
        assert got == GOLD["area"][f"tsv|{measure}|{thres}|{flank}"]


def test_cli_surface_matches_reference():
    from ld_tools_amd import cli
    tri = vars(cli.triangle_parser().parse_args(["-S", "s", "-D", "d"]))
    assert tri == {"src_dir_path": "s", "intgen_dir_path": "d", "trg_top_dir_path": None, "meta_lines_quan": 0,
                   "skip_intgen_data_ver": False, "gend_names": "both", "pop_names": "all", "ld_measure": "r_square",
                   "ld_low_thres": None, "matrix_type": "heatmap", "heatmap_json": False, "disp_letters": False,
                   "color_pal": "greens", "font_size": None, "square_shape": False, "dont_disp_footer": False,
                   "max_proc_quan": 4}                      # cli/ld_triangle_cli_en.py:40-74
    area = vars(cli.area_parser().parse_args(["-S", "s", "-D", "d"]))
    assert area == {"src_dir_path": "s", "intgen_dir_path": "d", "trg_top_dir_path": None, "meta_lines_quan": 0,
                    "skip_intgen_data_ver": False, "gend_names": "both", "pop_names": "all", "flank_size": 100000,
                    "ld_thres_measure": "r_square", "ld_low_thres": 0.8, "trg_file_type": "tsv", "max_proc_quan": 4}
    lite = vars(cli.lite_parser().parse_args(["rs1", "rs2", "-D", "d", "-g", "female", "-e", "eur,amr"]))
    assert lite == {"rs_id_1": "rs1", "rs_id_2": "rs2", "intgen_dir_path": "d", "skip_intgen_data_ver": False,
                    "gend_names": "female", "pop_names": "eur,amr"}
    with pytest.raises(SystemExit):
        cli.area_parser().parse_args(["-o", "xml"])
    a = cli.triangle_parser().parse_args(["-e", "eur,amr", "-g", "male"])
    assert cli._names(a) == (("male",), ("EUR", "AMR"))      # ld_triangle.py:33-38


def test_cli_language_follows_the_locale_and_survives_none(monkeypatch):
    """ld_triangle.py:386-389 picks the Russian help when the locale starts with 'ru' and raises on an unset locale; the
    shells here pick the same way from LC_ALL / LC_MESSAGES / LANG / locale.getlocale() and fall back to English."""
    from ld_tools_amd import cli
    for var in ("LDX_LANG", "LC_ALL", "LC_MESSAGES", "LANG"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr("locale.getlocale", lambda *a: (None, None))
    assert cli._lang() == "en"
    monkeypatch.setenv("LANG", "ru_RU.UTF-8")
    assert cli._lang() == "ru"
    ru, en_dests = cli.area_parser(), None
    monkeypatch.setenv("LDX_LANG", "en")
    en = cli.area_parser()
    assert [a.dest for a in ru._actions] == [a.dest for a in en._actions]          # same surface in both languages
    assert [a.default for a in ru._actions] == [a.default for a in en._actions]
    assert any("порог" in (a.help or "") for a in ru._actions) and all("порог" not in (a.help or "") for a in en._actions)


def test_sample_and_source_lookups(tmp_path):
    from ld_tools_amd.backend.create_src_dict import create_src_dict
    from ld_tools_amd.backend.get_sample_names import get_sample_names
    from ld_tools_amd.drivers import NotInIntgenConvDbError, NotRsIdError, check_rs_id
    db = tmp_path / "conversion.db"
    with sqlite3.connect(db) as conn:
        conn.execute("CREATE TABLE samples (sample TEXT, pop TEXT, super_pop TEXT, gender TEXT)")
        conn.executemany("INSERT INTO samples VALUES (?, ?, ?, ?)",
                         [("HG1", "GBR", "EUR", "male"), ("HG2", "FIN", "EUR", "female"), ("HG3", "PEL", "AMR", "female"),
                          ("HG4", "YRI", "AFR", "male")])
        conn.execute("CREATE TABLE variants (CHROM TEXT, POS INTEGER, ID TEXT)")
        conn.executemany("INSERT INTO variants VALUES (?, ?, ?)", [("6", 100, "rs1"), ("6", 50, "rs2"), ("7", 5, "rs3")])
    assert get_sample_names(("male", "female"), ("ALL",), str(db)) == ["HG1", "HG2", "HG3", "HG4"]
    assert get_sample_names(("female",), ("EUR", "PEL"), str(db)) == ["HG2", "HG3"]
    assert get_sample_names(("male",), ("AFR",), str(db)) == ["HG4"]
    (tmp_path / "src.tsv").write_text("header rs999\nx\trs1\ty\nrs2 and rs3 on one line\nno id here\nz rs3\nrs77777\n")
    d = create_src_dict(str(tmp_path), "src.tsv", 1, str(db))
    assert {c: sorted(v) for c, v in d.items()} == {"6": [[50, "rs2"], [100, "rs1"]], "7": [[5, "rs3"]]}
    assert create_src_dict(str(tmp_path), "src.tsv", 6, str(db)) == {}
    with sqlite3.connect(db) as conn:
        cur = conn.cursor()
        assert tuple(check_rs_id("rs1", cur)) == ("6", 100)
        with pytest.raises(NotRsIdError):
            check_rs_id("esv5", cur)
        with pytest.raises(NotInIntgenConvDbError):
            check_rs_id("rs424242", cur)


def test_ingest_helpers(chrom6):
    from ld_tools_amd.drivers import RaggedGenotypesError, codes_matrix, find_record, sample_genotypes
    from ld_tools_amd.drivers.area import build_ucsc_header
    from ld_tools_amd.drivers.ingest import k_to_python
    vcf, names, _, _ = chrom6
    rec = vcf.records[3]
    g = sample_genotypes(rec, names)
    assert len(g) == 2 * (len(names) - 1)                     # sample 7 is in no record: skipped (KeyError path)
    assert g[:4] == list(rec.samples[names[0]]["GT"]) + list(rec.samples[names[1]]["GT"])
    assert find_record(vcf, "6", rec.pos, rec.id) is rec
    assert find_record(vcf, "6", rec.pos, "rs0") is None
    dup = vcf.records[20]                                     # shares its position with record 19
    assert find_record(vcf, "6", dup.pos, dup.id) is dup
    m = codes_matrix([[1, 0, None, 2], [0, 0, 1, 1.0]])
    assert m.dtype == np.int8 and m.tolist() == [[1, 0, 2, 2], [0, 0, 1, 1]]
    with pytest.raises(RaggedGenotypesError):
        codes_matrix([[1, 0], [1, 0, 1]])
    with pytest.raises(ZeroDivisionError):
        codes_matrix([[1, 0], []])
    assert build_ucsc_header("chr", "6") == 'chr="6"'
    assert build_ucsc_header("pops", ("EUR", "AMR")) == 'pops="EUR","AMR"'
    assert build_ucsc_header("each_flank", 100000) == "each_flank=100000"
    vals = k_to_python(np.array([0.8216, -0.0, 0.0, 1.0, 0.0003], dtype=np.float32))
    assert [repr(v) for v in vals] == ["0.8216", "0", "0.0", "1.0", "0.0003"]


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_triangle_driver_writes_reference_text(chrom6, tmp_path):
    from ld_tools_amd.drivers import create_matrix, triangle_matrix, write_triangle_table
    vcf, names, tri_rows, _ = chrom6
    for key, text in GOLD["triangle"].items():
        measure, thres = key.split("|")
        thres = None if thres == "None" else float(thres)
        shuffled = tri_rows[5:] + tri_rows[:5]                # the driver sorts by position (ld_triangle.py:88)
        m = triangle_matrix(vcf, "6", shuffled, names, measure, thres)
        p = tmp_path / f"t_{measure}_{thres}.tsv"
        write_triangle_table(str(p), m, measure, ("EUR", "AMR"), ("male", "female"))
        assert p.read_text() == text
        assert all(type(v) is int and v == 0 for r in range(len(tri_rows)) for v in m.ld_two_dim[r][r:])
    before = vcf.fetches
    triangle_matrix(vcf, "6", tri_rows, names)
    assert vcf.fetches - before == len(tri_rows)              # one fetch per variant, not two per pair
    # folder / file naming of create_matrix (ld_triangle.py:66-69,236,348); a one-variant chromosome is skipped
    written = create_matrix(lambda chrom: vcf, {"6": tri_rows, "7": tri_rows[:1]}, "my.table.tsv", str(tmp_path), names,
                            "d_prime", 0.3, "table", ("EUR", "AMR"), ("male", "female"))
    assert written == [str(tmp_path / "my.table_LD_matr" / "my.table_chr6_d.tsv")]
    assert Path(written[0]).read_text() == GOLD["triangle"]["d_prime|0.3"]


@pytest.mark.gpu
def test_streamed_triangle_table_equals_list_writer(chrom6, tmp_path):
    """The block-streaming writer (for matrices too large for Python lists) writes the same bytes, including D' > 1
    cells (a variant with missing calls) and the int-0 / 0.0 distinction."""
    from ld_tools_amd import PackedPanel, ld_triangle
    from ld_tools_amd.drivers import stream_triangle_table, triangle_matrix, write_triangle_table
    from ld_tools_amd.drivers.ingest import codes_matrix, find_record, sample_genotypes
    vcf, names, tri_rows, _ = chrom6
    rows = sorted(tri_rows + [[vcf.records[5].pos, vcf.records[5].id]], key=lambda r: r[0])   # record 5 carries code-2 calls
    for measure, thres in (("r_square", None), ("d_prime", 0.3)):
        m = triangle_matrix(vcf, "6", rows, names, measure, thres)
        a, b = tmp_path / f"a_{measure}.tsv", tmp_path / f"b_{measure}.tsv"
        write_triangle_table(str(a), m, measure, ("ALL",), ("male", "female"))
        panel = PackedPanel.from_codes(codes_matrix([sample_genotypes(find_record(vcf, "6", p, i), names) for p, i in rows]))
        stream_triangle_table(str(b), "6", m.rs_ids_srtd, m.poss_srtd, ld_triangle(panel), measure, thres, ("ALL",),
                              ("male", "female"), rows_per_block=5)
        assert a.read_text() == b.read_text()


@pytest.mark.gpu
@pytest.mark.parametrize("ftype", ["tsv", "json", "rsids"])
def test_area_driver_writes_reference_text(chrom6, tmp_path, ftype):
    from ld_tools_amd.drivers import get_inld_vars
    vcf, names, _, queries = chrom6
    for k, (measure, thres, flank) in enumerate(AREA_CASES):
        want = GOLD["area"][f"{ftype}|{measure}|{thres}|{flank}"]
        top = tmp_path / f"case{k}"
        top.mkdir()
        written = get_inld_vars(lambda chrom: vcf, {"6": queries}, "q.txt", str(top), names, flank, measure, thres, ftype,
                                ("ALL",), ("female",))
        got = {os.path.basename(p): Path(p).read_text() for p in written}
        assert sorted(got) == sorted(want)
        for name in want:
            assert got[name] == want[name], name
        assert all(os.path.dirname(p) == str(top / "q_in_LD" / "6") for p in written)
    with pytest.raises(FileExistsError):                      # os.makedirs without exist_ok (ld_area.py:121-123)
        get_inld_vars(lambda chrom: vcf, {"6": queries[:1]}, "q.txt", str(tmp_path / "case0"), names)


@pytest.fixture(scope="module")
def ragged6():
    """Mixed ploidy: from record 24 on every second sample is haploid, so genotype lists of two lengths meet."""
    vcf, names = fakevcf.make_chromosome(haploid_from=24)
    uniq = _rows(vcf)
    return vcf, names, uniq[8:30:2], uniq[::4]


def test_restated_loops_with_oracle_match_reference_golden_ragged(ragged6):
    """CPU: the oracle's calc_ld keeps the reference's zip semantics (n = the shorter list, allele counts over the full
    lists, calc_ld.py:30-44) -- checked against texts made with the reference itself."""
    vcf, names, tri_rows, queries = ragged6
    for key, text in GOLD["triangle_ragged"].items():
        measure, thres = key.split("|")
        thres = None if thres == "None" else float(thres)
        assert ref_loops.triangle_tsv(vcf, "6", tri_rows, names, measure, thres, ("EUR",), ("male", "female"),
                                      orc.calc_ld_lists) == text
    for key, want in GOLD["area_ragged"].items():
        ftype, measure, thres, flank = key.split("|")
        assert ref_loops.area_files(vcf, "6", queries, names, int(flank), measure, float(thres), ftype, ("ALL",),
                                    ("male", "female"), orc.calc_ld_lists) == want


@pytest.mark.gpu
def test_drivers_mixed_ploidy_match_reference_text(ragged6, tmp_path):
    """Genotype lists of different lengths in one table (chrX across the PAR boundary): the drivers fall back to
    drivers/ragged.py -- pair_counts on length-truncated panels + the mirror epilogue, in batches -- and write the
    reference's bytes (ld_triangle table; ld_area tsv / json incl. the per-pair var_2_alt_freq)."""
    from ld_tools_amd.backend.calc_ld import calc_ld
    from ld_tools_amd.drivers import get_inld_vars, triangle_matrix, write_triangle_table
    from ld_tools_amd.drivers.ingest import find_record, sample_genotypes
    from ld_tools_amd.drivers.ragged import ragged_pairs
    vcf, names, tri_rows, queries = ragged6
    for key, text in GOLD["triangle_ragged"].items():
        measure, thres = key.split("|")
        thres = None if thres == "None" else float(thres)
        m = triangle_matrix(vcf, "6", tri_rows, names, measure, thres)
        p = tmp_path / f"r_{measure}_{thres}.tsv"
        write_triangle_table(str(p), m, measure, ("EUR",), ("male", "female"))
        assert p.read_text() == text
    for k, (key, want) in enumerate(GOLD["area_ragged"].items()):
        ftype, measure, thres, flank = key.split("|")
        top = tmp_path / f"ragged{k}"
        top.mkdir()
        written = get_inld_vars(lambda chrom: vcf, {"6": queries}, "q.txt", str(top), names, int(flank), measure,
                                float(thres), ftype, ("ALL",), ("male", "female"))
        got = {os.path.basename(p): Path(p).read_text() for p in written}
        assert sorted(got) == sorted(want)
        for name in want:
            assert got[name] == want[name], name
    # the batched helper against the fused drop-in, pair by pair, both orders (lengths 80/60, 60/80, 60/60, 80/80)
    genos = [sample_genotypes(find_record(vcf, "6", pos, rid), names) for pos, rid in tri_rows]
    assert len({len(g) for g in genos}) == 2
    pairs = [(i, j) for i in range(len(genos)) for j in range(len(genos)) if i != j]
    for (i, j), v in zip(pairs, ragged_pairs(genos, pairs)):
        w = calc_ld(genos[i], genos[j])
        assert str(v) == str(w), (i, j)


@pytest.mark.gpu
def test_area_driver_against_oracle_wide_window(chrom6):
    """A window that spans the whole chromosome, every record a query (incl. the MULTI_ALLELIC, non-rs and long-REF
    ones): hits equal the restated reference loop with the oracle's calc_ld."""
    from ld_tools_amd.drivers import area_scan
    from ld_tools_amd.drivers.area import HEADER_ROW
    vcf, names, _, _ = chrom6
    rows = [[r.pos, r.id] for r in vcf.records if r.id != "rs9029"]      # the duplicated id is ambiguous as a query
    res = area_scan(vcf, "6", rows, names, 10 ** 6, "r_square", 0.2)
    want = ref_loops.area_files(vcf, "6", rows, names, 10 ** 6, "r_square", 0.2, "json", ("ALL",), ("male",),
                                orc.calc_ld_lists)
    got = {}
    for r in res:
        if r.hits:
            got[f"{r.query_id}_chr6_r_0.2.json"] = [dict(zip(HEADER_ROW, r.query_ann))] + [dict(zip(HEADER_ROW, h)) for h in r.hits]
    assert sorted(got) == sorted(want)
    for name, text in want.items():
        assert json.loads(text)[1:] == json.loads(json.dumps(got[name])), name


@pytest.mark.gpu
def test_ld_lite_table(chrom6):
    pytest.importorskip("tabulate")
    from tabulate import tabulate
    from ld_tools_amd.drivers import ld_lite_table
    vcf, names, tri_rows, _ = chrom6
    (p1, id1), (p2, id2) = tri_rows[2], tri_rows[9]
    r1 = next(r for r in vcf.records if r.id == id1)
    r2 = next(r for r in vcf.records if r.id == id2)
    vals = orc.calc_ld_lists(ref_loops._genotypes(r1, names), ref_loops._genotypes(r2, names))
    want = tabulate([["chrom", "6", "6"], ["hg38_pos", p1, p2], ["alleles", "A/G", "A/G"], ["type", "SNP", "SNP"],
                     ["alt_freq", vals["var_1_alt_freq"], vals["var_2_alt_freq"]]],
                    headers=[tabulate([["r2", vals["r_square"]], ["D'", vals["d_prime"]], ["abs_dist", abs(p1 - p2)]],
                                      tablefmt="fancy_grid", disable_numparse=True), f"\n\n\n{id1}", f"\n\n\n{id2}"],
                    tablefmt="fancy_grid")                    # ld_lite.py:148-159
    assert ld_lite_table(vcf, "6", id1, p1, id2, p2, names) == want


# ------------------------------------------------------------------------------------------ the three shells, end to end
def _intgen_folder(tmp_path, vcf, names, genders):
    """A prepared '1000 Genomes folder' as the shells expect it: conversion.db with the ``samples`` and ``variants`` tables
    (backend/prep_intgen_data.py builds them in the reference; same-rsID repeats are dropped there too)."""
    d = tmp_path / "intgen"
    d.mkdir()
    with sqlite3.connect(d / "conversion.db") as conn:
        conn.execute("CREATE TABLE samples (sample TEXT, pop TEXT, super_pop TEXT, gender TEXT)")
        conn.executemany("INSERT INTO samples VALUES (?, ?, ?, ?)",
                         [(n, "GBR" if k % 2 else "PEL", "EUR" if k % 2 else "AMR", genders[k % len(genders)])
                          for k, n in enumerate(names)])
        conn.execute("CREATE TABLE variants (CHROM TEXT, POS INTEGER, ID TEXT)")
        seen = set()
        for r in vcf.records:
            if r.id.startswith("rs") and ";" not in r.id and r.id not in seen:
                seen.add(r.id)
                conn.execute("INSERT INTO variants VALUES (?, ?, ?)", (r.chrom, r.pos, r.id))
    return d


def _run_shell(script, argv, cwd):
    env = dict(os.environ, LDX_VCF_OPENER="fakevcf:opener_factory",
               PYTHONPATH=os.pathsep.join([str(ROOT / "tests"), str(ROOT), os.environ.get("PYTHONPATH", "")]))
    r = __import__("subprocess").run([sys.executable, str(ROOT / script)] + argv, capture_output=True, text=True, timeout=600,
                                     env=env, cwd=str(cwd))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.gpu
def test_ld_triangle_shell_end_to_end(chrom6, tmp_path):
    """`python3 ld_triangle.py -S ... -D ... -f -o table ...` with the reference's flags writes the reference's folder tree
    and table text (ld_triangle.py:364-411, 52-360): one run per (measure, threshold) of the golden set."""
    vcf, names, tri_rows, _ = chrom6
    intgen = _intgen_folder(tmp_path, vcf, names, ["male", "female"])
    src = tmp_path / "src"
    src.mkdir()
    (src / "study.tsv").write_text("col_a\tcol_b\n" + "".join(f"x\t{rs}\tgene\n" for _, rs in tri_rows[::-1]))
    for key, text in GOLD["triangle"].items():
        measure, thres = key.split("|")
        trg = tmp_path / f"out_{measure}_{thres}"
        trg.mkdir()
        argv = ["-S", str(src), "-D", str(intgen), "-t", str(trg), "-m", "1", "-f", "-e", "eur,amr", "-l", measure, "-o", "table",
                "-p", "8"]
        if thres != "None":
            argv += ["-z", thres]
        out = _run_shell("ld_triangle.py", argv, tmp_path)
        assert "parallel computation time" in out
        tree = sorted(str(p.relative_to(trg)) for p in trg.rglob("*") if p.is_file())
        assert tree == [f"study_LD_matr/study_chr6_{measure[0]}.tsv"]
        assert (trg / tree[0]).read_text() == text


@pytest.mark.gpu
def test_shells_run_several_tables_on_parallel_workers(chrom6, tmp_path):
    """-p / --max-proc-quan (ld_triangle.py:390-411): the reference maps its tables over a process pool; here the workers are
    threads of one process with a HIP stream each.  Five tables through 4 workers produce, file by file, the bytes of a
    one-worker run; the worker count follows the reference's rule (min of -p, the number of tables, 8)."""
    from ld_tools_amd.cli import _proc_quan
    assert [_proc_quan(4, 2), _proc_quan(4, 9), _proc_quan(16, 9), _proc_quan(16, 3), _proc_quan(1, 5)] == [2, 4, 8, 3, 1]
    vcf, names, tri_rows, queries = chrom6
    intgen = _intgen_folder(tmp_path, vcf, names, ["male", "female"])
    src = tmp_path / "src"
    src.mkdir()
    for k in range(5):
        rows = tri_rows[k:k + 9] if k else tri_rows
        (src / f"table{k}.tsv").write_text("h\n" + "".join(f"{rs}\tx\n" for _, rs in rows))
    trees = {}
    for workers in ("1", "4"):
        for script, extra in (("ld_triangle.py", ["-l", "d_prime", "-z", "0.3", "-o", "table"]),
                              ("ld_area.py", ["-w", "2500", "-l", "r_square", "-z", "0.05", "-o", "tsv"])):
            trg = tmp_path / f"out_{script[:-3]}_{workers}"
            trg.mkdir()
            out = _run_shell(script, ["-S", str(src), "-D", str(intgen), "-t", str(trg), "-m", "1", "-f", "-e", "eur,amr",
                                      "-p", workers] + extra, tmp_path)
            assert f"one HIP stream each): {workers}" in out
            trees[(script, workers)] = {str(p.relative_to(trg)): p.read_text() for p in trg.rglob("*") if p.is_file()}
    for script in ("ld_triangle.py", "ld_area.py"):
        assert trees[(script, "1")] and trees[(script, "1")] == trees[(script, "4")]
    assert trees[("ld_triangle.py", "4")]["table0_LD_matr/table0_chr6_d.tsv"] == GOLD["triangle"]["d_prime|0.3"]


@pytest.mark.gpu
@pytest.mark.parametrize("ftype", ["tsv", "json", "rsids"])
def test_ld_area_shell_end_to_end(chrom6, tmp_path, ftype):
    """`python3 ld_area.py -S ... -D ... -f -w ... -z ... -o ...`: folder tree {table}_in_LD/{chrom}/ and every file's bytes
    (ld_area.py:296-342, 62-292)."""
    vcf, names, _, queries = chrom6
    intgen = _intgen_folder(tmp_path, vcf, names, ["female"])          # the golden header says gends="female"
    src = tmp_path / "src"
    src.mkdir()
    (src / "gwas_hits.txt").write_text("".join(f"{rs} p=1e-9\n" for _, rs in queries))
    for k, (measure, thres, flank) in enumerate(AREA_CASES):
        want = GOLD["area"][f"{ftype}|{measure}|{thres}|{flank}"]
        trg = tmp_path / f"out{k}"
        trg.mkdir()
        _run_shell("ld_area.py", ["-S", str(src), "-D", str(intgen), "-t", str(trg), "-f", "-g", "female", "-w", str(flank),
                                  "-l", measure, "-z", str(thres), "-o", ftype], tmp_path)
        got = {str(p.relative_to(trg)): p.read_text() for p in trg.rglob("*") if p.is_file()}
        assert sorted(got) == sorted(f"gwas_hits_in_LD/6/{name}" for name in want)
        for name, text in want.items():
            assert got[f"gwas_hits_in_LD/6/{name}"] == text, name


@pytest.mark.gpu
def test_ld_lite_shell_end_to_end(chrom6, tmp_path):
    """`python3 ld_lite.py rsA rsB -D ... -f` prints the reference's nested table (ld_lite.py:49-159); its error types for
    a non-rs id, an unknown id and two chromosomes."""
    pytest.importorskip("tabulate")
    from tabulate import tabulate
    vcf, names, tri_rows, _ = chrom6
    intgen = _intgen_folder(tmp_path, vcf, names, ["male", "female"])
    (p1, id1), (p2, id2) = tri_rows[2], tri_rows[9]
    r1 = next(r for r in vcf.records if r.id == id1)
    r2 = next(r for r in vcf.records if r.id == id2)
    vals = orc.calc_ld_lists(ref_loops._genotypes(r1, names), ref_loops._genotypes(r2, names))
    want = tabulate([["chrom", "6", "6"], ["hg38_pos", p1, p2], ["alleles", "A/G", "A/G"], ["type", "SNP", "SNP"],
                     ["alt_freq", vals["var_1_alt_freq"], vals["var_2_alt_freq"]]],
                    headers=[tabulate([["r2", vals["r_square"]], ["D'", vals["d_prime"]], ["abs_dist", abs(p1 - p2)]],
                                      tablefmt="fancy_grid", disable_numparse=True), f"\n\n\n{id1}", f"\n\n\n{id2}"],
                    tablefmt="fancy_grid")
    out = _run_shell("ld_lite.py", [id1, id2, "-D", str(intgen), "-f"], tmp_path)
    assert out.rstrip("\n") == want
    from ld_tools_amd import cli
    from ld_tools_amd.drivers import DifChrsError, NotInIntgenConvDbError, NotRsIdError
    cli.VCF_OPENER_FACTORY = fakevcf.opener_factory
    try:
        with pytest.raises(NotRsIdError):
            cli.ld_lite_main(["esv1", id2, "-D", str(intgen), "-f"])
        with pytest.raises(NotInIntgenConvDbError):
            cli.ld_lite_main([id1, "rs424242", "-D", str(intgen), "-f"])
        with sqlite3.connect(intgen / "conversion.db") as conn:
            conn.execute("INSERT INTO variants VALUES ('7', 5, 'rs777')")
        with pytest.raises(DifChrsError):
            cli.ld_lite_main([id1, "rs777", "-D", str(intgen), "-f"])
    finally:
        cli.VCF_OPENER_FACTORY = None


@pytest.mark.gpu
def test_area_scan_reads_one_window_per_cluster(chrom6):
    """Queries far apart are fetched, packed and scanned per cluster of overlapping windows, not over the span between the
    outermost two; the results equal the single-region scan."""
    from ld_tools_amd.drivers import area_scan
    from ld_tools_amd.drivers.area import _clusters
    vcf, names, _, queries = chrom6
    wide = area_scan(vcf, "6", queries, names, 10 ** 6, "r_square", 0.3)      # one cluster: the whole chromosome
    spans = []
    real_fetch = vcf.fetch
    def spy(chrom, start, end):
        if end - start > 2:                                                   # window reads, not the per-query lookups
            spans.append((start, end))
        return real_fetch(chrom, start, end)
    vcf.fetch = spy
    try:
        narrow = area_scan(vcf, "6", queries, names, 150, "r_square", 0.3)
    finally:
        del vcf.fetch
    recs = [next(r for r in vcf.records if r.id == rs and r.pos == pos) for pos, rs in queries]
    assert len(spans) == len(_clusters(recs, 150)) > 3
    assert max(e - s for s, e in spans) < 3000 < vcf.records[-1].pos - vcf.records[0].pos
    assert sum(len(r.hits) for r in wide) > sum(len(r.hits) for r in narrow) > 0
    # the narrow scan equals the reference's loop (restated, with the oracle's calc_ld) query by query
    ref = ref_loops.area_files(vcf, "6", queries, names, 150, "r_square", 0.3, "rsids", ("ALL",), ("male",), orc.calc_ld_lists)
    got_ids = {f"{r.query_id}_chr6_r_0.3.txt": [r.query_id] + [h[1] for h in r.hits] for r in narrow if r.hits}
    assert sorted(got_ids) == sorted(ref)
    for name, text in ref.items():
        assert text.splitlines()[2:] == got_ids[name], name
