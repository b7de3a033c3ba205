"""ld_area driver: for every query variant, the variants of its flanks in LD above a threshold
(ld_area.py:62-292)."""
from __future__ import annotations

import json
import os
import re
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from ..ops import ld_area
from ..panel import PackedPanel
from .ingest import RaggedGenotypesError, codes_matrix, find_record, sample_genotypes
from .ragged import ragged_pairs

HEADER_ROW = ["hg38_pos", "rsID", "ref", "alt", "type", "alt_freq", "r2", "D'", "dist"]   # ld_area.py:97-105
_RS = re.compile(r"rs\d+$")


def build_ucsc_header(header_key, header_val) -> str:
    """One ``key=value`` element of the '##' header line (format of ld_area.py:3-14): a string value goes in double
    quotes, a tuple becomes its elements, each in double quotes, joined by commas; any other value is printed as is."""
    def quoted(text):
        return '"' + str(text) + '"'

    if isinstance(header_val, tuple):
        shown = ",".join(quoted(item) for item in header_val)
    elif isinstance(header_val, str):
        shown = quoted(header_val)
    else:
        shown = str(header_val)
    return header_key + "=" + shown


@dataclass
class AreaQueryResult:
    """One query variant and its hits, as the reference's annotation lists (ld_area.py:190-196,261-269)."""

    query_id: str
    query_ann: list                       # [pos, id, ref, alts, VT, alt_freq, 'quer', 'quer', 'quer']
    hits: List[list] = field(default_factory=list)   # [pos, id, ref, alts, VT, var_2_alt_freq, r2, D', dist]


def _ann(rec) -> list:
    return [rec.pos, rec.id, rec.ref, ",".join(rec.alts), ",".join(rec.info["VT"])]


def _clusters(queries, flank_size: int):
    """Group the query records whose fetch windows [max(0, pos - flank), pos + flank] overlap or touch: every group is
    read, packed and scanned on its own, so a handful of rsIDs spread over a chromosome costs a handful of windows, not the
    span between the outermost two (the reference reads one window per query, ld_area.py:174-217).  Returns
    [(lo, hi, [indices into ``queries``])] in ascending order of lo."""
    order = sorted(range(len(queries)), key=lambda k: queries[k].pos)
    groups = []
    for k in order:
        lo, hi = max(0, queries[k].pos - flank_size), queries[k].pos + flank_size
        if groups and lo <= groups[-1][1]:
            groups[-1][1] = max(groups[-1][1], hi)
            groups[-1][2].append(k)
        else:
            groups.append([lo, hi, [k]])
    return [(g[0], g[1], g[2]) for g in groups]


def area_scan(vcf, chrom, chrom_rows: Sequence[Sequence], sample_names: Sequence[str], flank_size: int = 100000,
              ld_thres_measure: str = "r_square", ld_low_thres: float = 0.8) -> List[AreaQueryResult]:
    """The window loop of ld_area.py:152-276 for one chromosome, in the order of ``chrom_rows``.

    Reference semantics kept: the window is pysam's fetch(chrom, max(0, pos - flank), pos + flank), i.e. every
    record that overlaps that 0-based half-open interval (start = pos - 1, stop = start + len(ref)); opposing
    records with the query's id, an id that is not ``rs<digits>`` or a MULTI_ALLELIC flag are skipped
    (ld_area.py:222-225); var_1 = query, var_2 = opposing; a hit needs rounded measure >= threshold
    (ld_area.py:248); hits come in VCF order.  Batched per cluster of overlapping windows: the cluster's region is
    read once, packed once, and one windowed kernel launch evaluates every (query, opposing) pair in it."""
    queries = []
    for pos, rs_id in chrom_rows:
        rec = find_record(vcf, chrom, int(pos), rs_id)
        if rec is None:
            raise UnboundLocalError(f"no 1000 Genomes record for {rs_id} at {chrom}:{pos}")   # ld_area.py:160 uses an unbound name
        queries.append(rec)
    if not queries:
        return []
    key = lambda r: (r.pos, r.id, r.ref, tuple(r.alts))              # noqa: E731  identity of a record
    results: List[Optional[AreaQueryResult]] = [None] * len(queries)
    for lo, hi, members in _clusters(queries, flank_size):
        region = list(vcf.fetch(chrom, lo, hi))                          # VCF order
        index = {}
        for k, r in enumerate(region):
            index.setdefault(key(r), k)
        q_rows = {m: index[key(queries[m])] for m in members}           # every query lies in its cluster's region
        genotypes = [sample_genotypes(r, sample_names) for r in region]
        positions = np.array([r.pos for r in region], dtype=np.int64)
        stops = np.array([r.pos - 1 + len(r.ref) for r in region], dtype=np.int64)
        eligible = np.array([_RS.match(r.id or "") is not None and "MULTI_ALLELIC" not in r.info for r in region])
        try:
            codes = codes_matrix(genotypes)
        except RaggedGenotypesError:
            # mixed ploidy inside this cluster: the window and the filters applied on the host, calc_ld in batches by
            # pairs of genotype-list lengths (drivers/ragged.py: zip semantics of calc_ld.py:30-31)
            for m in members:
                q, qrow = queries[m], q_rows[m]
                low, high = max(0, q.pos - flank_size), q.pos + flank_size
                qg = genotypes[qrow]
                res = AreaQueryResult(q.id, _ann(q) + [round(list(qg).count(1) / len(qg), 4)] + ["quer"] * 3)   # ld_area.py:188-196
                opp = [k for k, o in enumerate(region)
                       if eligible[k] and o.id != q.id and positions[k] - 1 < high and stops[k] > low]
                for orow, v in zip(opp, ragged_pairs(genotypes, [(qrow, k) for k in opp])):
                    if v[ld_thres_measure] < ld_low_thres:            # ld_area.py:248
                        continue
                    o = region[orow]
                    res.hits.append(_ann(o) + [v["var_2_alt_freq"], v["r_square"], v["d_prime"], o.pos - q.pos])
                results[m] = res
            continue
        panel = PackedPanel.from_codes(codes)
        # the kernel's window is positional (low < pos_o <= high); long REF alleles that start before the window but
        # overlap it are reached by widening the lower flank, and the exact overlap rule is applied to the hits below
        extra = int((stops - (positions - 1)).max()) - 1
        uniq_q = sorted(set(q_rows.values()))
        hits = ld_area(panel, positions, uniq_q, flank=flank_size + extra, measure=ld_thres_measure, thres=ld_low_thres)
        hq = hits.query.cpu().numpy()
        ho = hits.oppos.cpu().numpy()
        hv = hits.python_values(panel)                                   # [(r2, D')] as the reference's Python values
        alt_freq = panel.alt_freq4().cpu().numpy()
        by_query = {}
        for qrow, orow, (r2, dp) in zip(hq.tolist(), ho.tolist(), hv):
            by_query.setdefault(qrow, []).append((orow, r2, dp))
        for m in members:
            q, qrow = queries[m], q_rows[m]
            low = max(0, q.pos - flank_size)
            high = q.pos + flank_size
            res = AreaQueryResult(q.id, _ann(q) + [float(alt_freq[qrow])] + ["quer"] * 3)    # ld_area.py:188-196
            for orow, r2, dp in by_query.get(qrow, ()):                 # ascending panel row = VCF order
                o = region[orow]
                if not eligible[orow] or o.id == q.id:
                    continue
                if not (positions[orow] - 1 < high and stops[orow] > low):                # pysam overlap with [low, high)
                    continue
                res.hits.append(_ann(o) + [float(alt_freq[orow]), r2, dp, o.pos - q.pos])
            results[m] = res
    return results


def write_area_file(path: str, res: AreaQueryResult, trg_file_type: str, ucsc_header_line: str,
                    meta_keys: Sequence[str], meta_vals: Sequence) -> bool:
    """One result file in the reference's three formats (ld_area.py:200-211,261-289).  Returns False -- and writes
    nothing -- when the query has no hit: the reference creates the file and removes it again (ld_area.py:291-292)."""
    if not res.hits:
        return False
    with open(path, "w") as out:
        if trg_file_type == "rsids":
            out.write(ucsc_header_line + "\n")
            out.write("#rsID\n")
            out.write(res.query_id + "\n")
            for hit in res.hits:
                out.write(hit[1] + "\n")
        elif trg_file_type == "tsv":
            out.write(ucsc_header_line + "\n")
            out.write("#" + "\t".join(HEADER_ROW) + "\n")
            out.write("\t".join(map(str, res.query_ann)) + "\n")
            for hit in res.hits:
                out.write("\t".join(map(str, hit)) + "\n")
        elif trg_file_type == "json":
            trg_obj = [dict(zip(meta_keys, meta_vals)), dict(zip(HEADER_ROW, res.query_ann))]
            trg_obj += [dict(zip(HEADER_ROW, hit)) for hit in res.hits]
            json.dump(trg_obj, out, indent=4)
        else:
            raise ValueError(f"unknown trg_file_type {trg_file_type!r}")
    return True


def get_inld_vars(vcf_opener, data_by_chrs: dict, src_file_name: str, trg_top_dir_path: str,
                  sample_names: Sequence[str], flank_size: int = 100000, ld_thres_measure: str = "r_square",
                  ld_low_thres: float = 0.8, trg_file_type: str = "tsv", pop_names: Sequence[str] = ("ALL",),
                  gend_names: Sequence[str] = ("male", "female")) -> List[str]:
    """PrepSingleProc.get_inld_vars (ld_area.py:62-292) for one source table: folder
    ``{src_file_base}_in_LD/{chrom}/`` with one file ``{rsID}_chr{chrom}_{m}_{thres}.{ext}`` per query that has at
    least one hit.  ``os.makedirs`` without exist_ok, like the reference (ld_area.py:121-123).  Returns the paths."""
    src_file_base = src_file_name.rsplit(".", maxsplit=1)[0]
    trg_dir_path = os.path.join(trg_top_dir_path, f"{src_file_base}_in_LD")
    ext = trg_file_type if trg_file_type in ("tsv", "json") else "txt"
    meta_keys = ["chr", "gends", "pops", "each_flank", f"{ld_thres_measure}_thres"]
    written = []
    for chrom in data_by_chrs:
        chr_dir_path = os.path.join(trg_dir_path, chrom)
        os.makedirs(chr_dir_path)
        meta_vals = [chrom, tuple(gend_names), tuple(pop_names), flank_size, ld_low_thres]
        ucsc_header_line = "##" + " ".join(map(build_ucsc_header, meta_keys, meta_vals))
        vcf = vcf_opener(chrom)
        try:
            results = area_scan(vcf, chrom, data_by_chrs[chrom], sample_names, flank_size, ld_thres_measure,
                                ld_low_thres)
        finally:
            close = getattr(vcf, "close", None)
            if close:
                close()
        for res in results:
            name = f"{res.query_id}_chr{chrom}_{ld_thres_measure[0]}_{str(ld_low_thres)}.{ext}"
            path = os.path.join(chr_dir_path, name)
            if write_area_file(path, res, trg_file_type, ucsc_header_line, meta_keys, meta_vals):
                written.append(path)
    return written
