"""An in-memory stand-in for pysam.VariantFile / VariantRecord, and a deterministic synthetic chromosome.

Only what the reference's drivers touch is modelled (ld_triangle.py:160-186, ld_area.py:153-235, ld_lite.py:109-137):
``fetch(chrom, start, end)`` returning the records that overlap the 0-based half-open interval in file order, and
records with ``id, pos, ref, alts, info, samples[name]['GT']``.  Test infrastructure only.
"""
from __future__ import annotations

import numpy as np

from ld_tools_amd import synth


class FakeRecord:
    def __init__(self, chrom, pos, rs_id, ref, alts, info, samples):
        self.chrom, self.pos, self.id, self.ref, self.alts, self.info, self.samples = chrom, pos, rs_id, ref, alts, info, samples
        self.start = pos - 1
        self.stop = self.start + len(ref)


class FakeVcf:
    def __init__(self, records):
        self.records = list(records)          # file order: ascending start, ties in insertion order
        self.fetches = 0

    def fetch(self, chrom, start, end):
        self.fetches += 1
        for r in self.records:
            if r.chrom == chrom and r.start < end and r.stop > start:
                yield r

    def close(self):
        pass


def make_chromosome(chrom="6", n_variants=48, n_samples=40, seed=11, first_pos=1000, step=137, haploid_from=None):
    """Records with LD blocks (ld_tools_amd.synth), plus the oddities the reference's filters and genotype
    assembly react to: a MULTI_ALLELIC record, a non-rs id, a duplicated rsID, a long deletion (REF of 40 bases),
    two records at one position, a sample absent from some records, missing (None) and second-ALT (2) calls.
    ``haploid_from``: records from that index on carry one-allele GTs for every second sample, so genotype lists of
    two lengths meet (the reference zips them, calc_ld.py:30-31)."""
    codes = synth.synth_codes_host(n_variants, 2 * n_samples, seed=seed)
    names = [f"HG{100 + s:05d}" for s in range(n_samples)]
    rng = np.random.RandomState(seed)
    records = []
    pos = first_pos
    for v in range(n_variants):
        pos += step if v % 7 else 3            # some close neighbours
        if v == 20:
            pos = records[-1].pos              # two records at the same position
        rs_id = f"rs{9000 + v}"
        ref, alts, info = "A", ("G",), {"VT": ("SNP",)}
        if v == 5:
            info = {"VT": ("SNP",), "MULTI_ALLELIC": True}
            alts = ("G", "T")
        if v == 9:
            rs_id = "esv123456"
        if v == 13:
            rs_id = "rs9012;rs77"              # not rs<digits>$ for the opposing filter
        if v == 17:
            ref, info = "ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT", {"VT": ("INDEL",)}
        if v == 30:
            rs_id = "rs9029"                   # duplicates the id of variant 29
        samples = {}
        for s, name in enumerate(names):
            if v % 11 == 3 and s == 7:
                continue                        # this record does not carry sample 7 ... nor any other record: see below
            gt = [int(codes[v, 2 * s]), int(codes[v, 2 * s + 1])]
            if rng.rand() < 0.01:
                gt[rng.randint(2)] = None
            if v == 5 and rng.rand() < 0.2:
                gt[0] = 2
            if haploid_from is not None and v >= haploid_from and s % 2 == 1:
                gt = gt[:1]                     # mixed ploidy (chrX past the PAR boundary): every second sample is haploid
            samples[name] = {"GT": tuple(gt)}
        records.append(FakeRecord(chrom, pos, rs_id, ref, alts, info, samples))
    # drop sample 7 from every record and keep it in sample_names, so the KeyError-skip path (ld_triangle.py:170-171)
    # is exercised on every record (genotype lists of different lengths are what ``haploid_from`` is for)
    for r in records:
        r.samples.pop(names[7], None)
    return FakeVcf(records), names


def opener_factory(intgen_dir_path):
    """For ld_tools_amd.cli (LDX_VCF_OPENER="fakevcf:opener_factory"): every chromosome opens the synthetic one."""
    vcf, _ = make_chromosome()
    return lambda chrom: vcf
