"""ld_triangle driver: one LD matrix per chromosome (ld_triangle.py:52-360 without the plotly heat map)."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from .._lib import ONE_MEASURE_FMT
from ..ops import ld_triangle
from ..panel import PackedPanel
from .ingest import RaggedGenotypesError, codes_matrix, find_record, k_to_python, sample_genotypes
from .ragged import ragged_pairs


@dataclass
class TriangleMatrix:
    """What the reference holds for one chromosome after its pair loop (ld_triangle.py:88-230)."""

    chrom: str
    rs_ids_srtd: List[str]
    poss_srtd: List[int]
    ld_two_dim: List[list]            # lower triangle: rounded measure (float, or int 0); everything else int 0
    alleles: List[str]                # ref + '/' + alts[0] per variant (ld_triangle.py:165,179)
    types: List[str]                  # info['VT'][0] per variant      (ld_triangle.py:166,180)
    alt_freqs: List[float]            # round(a / n, 4) per variant     (calc_ld.py:96-97)


def triangle_matrix(vcf, chrom, chrom_rows: Sequence[Sequence], sample_names: Sequence[str],
                    ld_measure: str = "r_square", ld_low_thres: Optional[float] = None) -> TriangleMatrix:
    """The pair loop of ld_triangle.py:133-230 for one chromosome: every row > col pair of the position-sorted
    variants, var_1 = row (larger position), var_2 = col; cells whose rounded measure is below ``ld_low_thres``
    keep the template's int 0 (ld_triangle.py:223-225).  Each variant is fetched and packed ONCE."""
    rows = sorted(chrom_rows, key=lambda r: r[0])                     # ld_triangle.py:88 (stable, by position)
    poss = [int(r[0]) for r in rows]
    rs_ids = [r[1] for r in rows]
    genotypes, alleles, types = [], [], []
    for pos, rs_id in zip(poss, rs_ids):
        rec = find_record(vcf, chrom, pos, rs_id)
        # no matching record: the reference's genotype list stays empty and calc_ld divides by zero (calc_ld.py:33)
        genotypes.append(sample_genotypes(rec, sample_names) if rec is not None else [])
        alleles.append(rec.ref + "/" + rec.alts[0] if rec is not None else "")
        types.append(rec.info["VT"][0] if rec is not None else "")
    n = len(rows)
    try:
        codes = codes_matrix(genotypes)
    except RaggedGenotypesError:
        # mixed ploidy: genotype lists of different lengths.  The reference zips them pair by pair (calc_ld.py:30-31);
        # drivers/ragged.py does the same in batches, one pair of length groups at a time
        pairs = [(r, c) for r in range(n) for c in range(r)]
        vals = ragged_pairs(genotypes, pairs)
        ld_two_dim = [[0] * n for _ in range(n)]                      # the template of int zeros (ld_triangle.py:133-141)
        for (r, c), v in zip(pairs, vals):
            if ld_low_thres is not None and v[ld_measure] < ld_low_thres:
                continue                                              # ld_triangle.py:223-225
            ld_two_dim[r][c] = v[ld_measure]
        alt_freqs = [round(list(g).count(1) / len(g), 4) for g in genotypes]
        return TriangleMatrix(chrom, rs_ids, poss, ld_two_dim, alleles, types, alt_freqs)
    panel = PackedPanel.from_codes(codes)
    # the table holds ONE measure (ld_triangle.py:223-230): 2-byte cells -- k and the int-0 mark of that measure, lossless;
    # the kernel skips the other value's arithmetic
    res = ld_triangle(panel, fmt=ONE_MEASURE_FMT[ld_measure])
    dense, fixes = res.dense_values(ld_measure, ld_low_thres)        # -0.0 = the template's / a computed int 0
    flat = k_to_python(dense, fixes)
    ld_two_dim = [flat[r * n:(r + 1) * n] for r in range(n)]
    alt_freqs = panel.alt_freq4().cpu().numpy().tolist()
    return TriangleMatrix(chrom, rs_ids, poss, ld_two_dim, alleles, types, alt_freqs)


def write_triangle_table(path: str, m: TriangleMatrix, ld_measure: str, pop_names: Sequence[str],
                         gend_names: Sequence[str]) -> None:
    """The tabular matrix file, byte for byte (ld_triangle.py:344-360): general-info header, an empty line, the
    rsID and position header rows, then one row per variant with str() of every cell ('0' for the int 0)."""
    tab = "\t"
    poss = [str(p) for p in m.poss_srtd]
    with open(path, "w") as out:
        out.write(f"##General\tinfo:\t{ld_measure}\tchr{m.chrom}\t{tab.join(pop_names)}\t{tab.join(gend_names)}\n\n")
        out.write("rsIDs\t\t" + "\t".join(m.rs_ids_srtd) + "\n")
        out.write("\tPositions\t" + "\t".join(poss) + "\n")
        for row_index in range(len(m.rs_ids_srtd)):
            line = "\t".join(map(str, m.ld_two_dim[row_index])) + "\n"
            out.write(m.rs_ids_srtd[row_index] + "\t" + poss[row_index] + "\t" + line)


_K_TEXT = None


def _cell_text_table() -> np.ndarray:
    """str() of every value a cell can hold with k = value * 10^4 in 0..10000: index k -> repr(k / 10^4), and one more
    entry (index 10001) for the int 0."""
    global _K_TEXT
    if _K_TEXT is None:
        _K_TEXT = np.array([str(k / 10000.0) for k in range(10001)] + ["0"], dtype=object)
    return _K_TEXT


def stream_triangle_table(path: str, chrom: str, rs_ids_srtd: Sequence[str], poss_srtd: Sequence[int], result,
                          ld_measure: str, ld_low_thres: Optional[float], pop_names: Sequence[str],
                          gend_names: Sequence[str], rows_per_block: int = 512) -> None:
    """write_triangle_table for matrices too large to hold as Python lists: the same bytes, produced block of rows
    by block of rows straight from the device result (``TriangleResult.dense(rows=...)``) through a 10 002-entry
    text table instead of one Python float per cell.  A 10 000 x 10 000 matrix (~600 MB of text, which the
    reference could never produce) takes tens of seconds, almost all of it string joining."""
    tab = "\t"
    poss = [str(p) for p in poss_srtd]
    n = len(rs_ids_srtd)
    table = _cell_text_table()
    with open(path, "w") as out:
        out.write(f"##General\tinfo:\t{ld_measure}\tchr{chrom}\t{tab.join(pop_names)}\t{tab.join(gend_names)}\n\n")
        out.write("rsIDs\t\t" + "\t".join(rs_ids_srtd) + "\n")
        out.write("\tPositions\t" + "\t".join(poss) + "\n")
        for r0 in range(0, n, rows_per_block):
            r1 = min(n, r0 + rows_per_block)
            v, fixes = result.dense_values(ld_measure, ld_low_thres, rows=(r0, r1))
            esc = np.isnan(v)
            k = np.rint(np.where(esc, 0, v).astype(np.float64) * 1e4).astype(np.int64)
            int0 = np.signbit(v) & (v == 0)
            big = (k > 10000) & ~esc                          # D' or r^2 above 1: only with missing codes; rare, formatted one by one
            idx = np.where(int0, 10001, np.minimum(k, 10000))
            text = table[idx]
            if big.any():
                for rr, cc in zip(*np.nonzero(big)):
                    text[rr, cc] = str(k[rr, cc] / 10000.0)
            for (ar, ac), val in fixes.items():               # values the cell format cannot hold: exact, from ld_pairs
                text[ar - r0, ac] = str(val)
            for row_index in range(r0, r1):
                out.write(rs_ids_srtd[row_index] + "\t" + poss[row_index] + "\t" + "\t".join(text[row_index - r0]) + "\n")


def create_matrix(vcf_opener, data_by_chrs: dict, src_file_name: str, trg_top_dir_path: str,
                  sample_names: Sequence[str], ld_measure: str = "r_square", ld_low_thres: Optional[float] = None,
                  matrix_type: str = "table", pop_names: Sequence[str] = ("ALL",),
                  gend_names: Sequence[str] = ("male", "female")) -> List[str]:
    """PrepSingleProc.create_matrix (ld_triangle.py:52-360) for one source table: a sub-folder
    ``{src_file_base}_LD_matr`` with one ``{src_file_base}_chr{chrom}_{m}.tsv`` per chromosome that has at least
    two variants (ld_triangle.py:80-83,236,348).  ``vcf_opener(chrom)`` returns the opened VCF of a chromosome
    (the reference opens ``{chrom}.vcf.gz`` under the 1000 Genomes folder, ld_triangle.py:128-129).  Heat maps
    (plotly, ld_triangle.py:239-340) are outside this package: ``matrix_type`` 'heatmap' / 'both' produce the
    table only.  Returns the paths written."""
    src_file_base = src_file_name.rsplit(".", maxsplit=1)[0]
    trg_dir_path = os.path.join(trg_top_dir_path, f"{src_file_base}_LD_matr")
    written = []
    for chrom in data_by_chrs:
        if len(data_by_chrs[chrom]) < 2:
            continue
        if not os.path.exists(trg_dir_path):
            os.mkdir(trg_dir_path)
        vcf = vcf_opener(chrom)
        try:
            m = triangle_matrix(vcf, chrom, data_by_chrs[chrom], sample_names, ld_measure, ld_low_thres)
        finally:
            close = getattr(vcf, "close", None)
            if close:
                close()
        trg_file_base = f"{src_file_base}_chr{chrom}_{ld_measure[0]}"
        path = os.path.join(trg_dir_path, trg_file_base + ".tsv")
        write_triangle_table(path, m, ld_measure, pop_names, gend_names)
        written.append(path)
    return written
