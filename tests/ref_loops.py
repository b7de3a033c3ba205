"""The reference's driver loops and writers restated on top of a ``calc_ld`` callable -- test infrastructure.

``calc`` is the oracle's calc_ld_lists at test time, and the reference's own calc_ld when
tests/golden/make_golden_drivers.py builds the committed expectations.  Every block cites the lines it follows;
unlike the product drivers these re-fetch and re-assemble genotypes per pair exactly as the reference does.
"""
from __future__ import annotations

import io
import json
import re


def _genotypes(rec, sample_names):                 # ld_triangle.py:167-171
    out = []
    for name in sample_names:
        try:
            out += rec.samples[name]["GT"]
        except KeyError:
            continue
    return out


def triangle_tsv(vcf, chrom, chrom_rows, sample_names, ld_measure, ld_low_thres, pop_names, gend_names, calc) -> str:
    rows = sorted(chrom_rows, key=lambda row: row[0])                       # ld_triangle.py:88
    poss_srtd = [r[0] for r in rows]
    rs_ids_srtd = [r[1] for r in rows]
    n = len(rows)
    ld_two_dim = [[0 for _ in range(n)] for _ in range(n)]                  # :114
    for row_index in range(n):                                             # :133-230
        for col_index in range(n):
            if row_index <= col_index:
                continue
            y, x = [], []
            for rec in vcf.fetch(chrom, rows[row_index][0] - 1, rows[row_index][0]):
                if rec.id != rows[row_index][1]:
                    continue
                y = _genotypes(rec, sample_names)
                break
            for rec in vcf.fetch(chrom, rows[col_index][0] - 1, rows[col_index][0]):
                if rec.id != rows[col_index][1]:
                    continue
                x = _genotypes(rec, sample_names)
                break
            vals = calc(y, x)                                               # :193-194  var_1 = row, var_2 = col
            if ld_low_thres is not None and vals[ld_measure] < ld_low_thres:
                continue
            ld_two_dim[row_index][col_index] = vals[ld_measure]
    out = io.StringIO()                                                     # :353-360
    tab, poss = "\t", [str(p) for p in poss_srtd]
    out.write(f"##General\tinfo:\t{ld_measure}\tchr{chrom}\t{tab.join(pop_names)}\t{tab.join(gend_names)}\n\n")
    out.write("rsIDs\t\t" + "\t".join(rs_ids_srtd) + "\n")
    out.write("\tPositions\t" + "\t".join(poss) + "\n")
    for row_index in range(n):
        line = "\t".join(map(str, ld_two_dim[row_index])) + "\n"
        out.write(rs_ids_srtd[row_index] + "\t" + poss[row_index] + "\t" + line)
    return out.getvalue()


def _ucsc(key, val):                                                        # ld_area.py:3-14
    if type(val).__name__ == "str":
        val = f'"{val}"'
    elif type(val).__name__ == "tuple":
        val = ",".join([f'"{e}"' for e in val])
    return f"{key}={val}"


def area_files(vcf, chrom, chrom_rows, sample_names, flank_size, ld_thres_measure, ld_low_thres, trg_file_type,
               pop_names, gend_names, calc) -> dict:
    """{file name: text} for the queries with at least one hit (ld_area.py:86-292)."""
    ext = trg_file_type if trg_file_type in ("tsv", "json") else "txt"
    meta_keys = ["chr", "gends", "pops", "each_flank", f"{ld_thres_measure}_thres"]
    header_row = ["hg38_pos", "rsID", "ref", "alt", "type", "alt_freq", "r2", "D'", "dist"]
    meta_vals = [chrom, tuple(gend_names), tuple(pop_names), flank_size, ld_low_thres]
    ucsc_header_line = "##" + " ".join(map(_ucsc, meta_keys, meta_vals))
    files = {}
    for var_row in chrom_rows:
        for rec in vcf.fetch(chrom, var_row[0] - 1, var_row[0]):            # :153-159
            if rec.id != var_row[1]:
                continue
            query = rec
            break
        name = f"{query.id}_chr{chrom}_{ld_thres_measure[0]}_{str(ld_low_thres)}.{ext}"
        empty_res = True
        low_bound = max(0, query.pos - flank_size)                          # :174-177
        high_bound = query.pos + flank_size
        qg = _genotypes(query, sample_names)
        q_alt_freq = round(qg.count(1) / len(qg), 4)                        # :188-189
        q_ann = [query.pos, query.id, query.ref, ",".join(query.alts), ",".join(query.info["VT"]), q_alt_freq] + ["quer"] * 3
        out = io.StringIO()
        if trg_file_type == "rsids":
            out.write(ucsc_header_line + "\n#rsID\n" + query.id + "\n")
        elif trg_file_type == "tsv":
            out.write(ucsc_header_line + "\n#" + "\t".join(header_row) + "\n" + "\t".join(map(str, q_ann)) + "\n")
        else:
            trg_obj = [dict(zip(meta_keys, meta_vals)), dict(zip(header_row, q_ann))]
        for opp in vcf.fetch(chrom, low_bound, high_bound):                 # :215-276
            if opp.id == query.id or re.match(r"rs\d+$", opp.id) is None or "MULTI_ALLELIC" in opp.info:
                continue
            vals = calc(qg, _genotypes(opp, sample_names))
            if vals[ld_thres_measure] < ld_low_thres:
                continue
            empty_res = False
            if trg_file_type == "rsids":
                out.write(opp.id + "\n")
                continue
            ann = [opp.pos, opp.id, opp.ref, ",".join(opp.alts), ",".join(opp.info["VT"]), vals["var_2_alt_freq"],
                   vals["r_square"], vals["d_prime"], opp.pos - query.pos]
            if trg_file_type == "tsv":
                out.write("\t".join(map(str, ann)) + "\n")
            else:
                trg_obj.append(dict(zip(header_row, ann)))
        if trg_file_type == "json":
            json.dump(trg_obj, out, indent=4)
        if not empty_res:                                                    # :291-292
            files[name] = out.getvalue()
    return files
