"""ld_lite: LD of one pair of rsIDs as a printed table (ld_lite.py:1-159)."""
from __future__ import annotations

import re

from ..backend.calc_ld import calc_ld
from .ingest import sample_genotypes


class NotRsIdError(Exception):
    """ld_lite.py:3-10"""

    def __init__(self, rs_id):
        super().__init__(f"{rs_id} is non-rs identifier")


class NotInIntgenConvDbError(Exception):
    """ld_lite.py:12-20"""

    def __init__(self, rs_id):
        super().__init__(f"{rs_id} is not available in 1000 Genomes")


class DifChrsError(Exception):
    """ld_lite.py:22-31"""

    def __init__(self, rs_id_1, rs_id_2):
        super().__init__(f"{rs_id_1} and {rs_id_2} belong to different chromosomes")


def check_rs_id(rs_id, cursor):
    """Validate an identifier and look up (CHROM, POS) in the ``variants`` table (ld_lite.py:33-45).  The id is
    bound as a parameter instead of being pasted into the SQL text."""
    if re.search(r"rs\d+\b", rs_id) is None:
        raise NotRsIdError(rs_id)
    cursor.execute("SELECT CHROM, POS FROM variants WHERE ID = ?", (rs_id,))
    var_basic_info = cursor.fetchone()
    if var_basic_info is None:
        raise NotInIntgenConvDbError(rs_id)
    return var_basic_info


def ld_lite_table(vcf, chrom, rs_id_1: str, var_1_pos: int, rs_id_2: str, var_2_pos: int, sample_names) -> str:
    """The nested fancy_grid table of ld_lite.py:148-159 for two variants of one chromosome (tabulate needed)."""
    from tabulate import tabulate

    found = []
    for rs_id, pos in ((rs_id_1, var_1_pos), (rs_id_2, var_2_pos)):
        genotypes, alleles, var_type = [], None, None
        for rec in vcf.fetch(chrom, pos - 1, pos):
            if rec.id != rs_id:
                continue
            alleles = rec.ref + "/" + rec.alts[0]
            var_type = rec.info["VT"][0]
            genotypes = sample_genotypes(rec, sample_names)
            break
        found.append((genotypes, alleles, var_type))
    (g1, var_1_alleles, var_1_type), (g2, var_2_alleles, var_2_type) = found
    trg_vals = calc_ld(g1, g2)
    return tabulate([["chrom", chrom, chrom],
                     ["hg38_pos", var_1_pos, var_2_pos],
                     ["alleles", var_1_alleles, var_2_alleles],
                     ["type", var_1_type, var_2_type],
                     ["alt_freq", trg_vals["var_1_alt_freq"], trg_vals["var_2_alt_freq"]]],
                    headers=[tabulate([["r2", trg_vals["r_square"]],
                                       ["D'", trg_vals["d_prime"]],
                                       ["abs_dist", abs(var_1_pos - var_2_pos)]],
                                      tablefmt="fancy_grid", disable_numparse=True),
                             f"\n\n\n{rs_id_1}", f"\n\n\n{rs_id_2}"],
                    tablefmt="fancy_grid")
