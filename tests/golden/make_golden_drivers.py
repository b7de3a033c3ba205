#!/usr/bin/env python3
"""Expected driver outputs (text) made with the REAL reference calc_ld -> tests/golden/driver_text.json.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_drivers.py [--out DIR]
The reference is loaded by file path (tests/golden/_reference.py), never through sys.path.
The loops and writers are tests/ref_loops.py (a restatement); only calc_ld comes from the reference.  Stored:
the texts, not the reference's source.
"""
import json
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.dont_write_bytecode = True
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import _reference  # noqa: E402

ref_calc_ld = _reference.load_reference_calc_ld()      # the reference, checked by code-object file name
OUT = Path(sys.argv[sys.argv.index("--out") + 1]) if "--out" in sys.argv else HERE
OUT.mkdir(parents=True, exist_ok=True)

import fakevcf  # noqa: E402
import ref_loops  # noqa: E402

vcf, names = fakevcf.make_chromosome()
rows = [[r.pos, r.id] for r in vcf.records if r.id.startswith("rs") and ";" not in r.id]
seen, uniq = set(), []
for r in rows:                       # one row per rsID (the conversion database holds each id once)
    if r[1] not in seen:
        seen.add(r[1])
        uniq.append(r)
tri_rows = uniq[::2][:16]
out = {"triangle": {}, "area": {}}
for measure in ("r_square", "d_prime"):
    for thres in (None, 0.3):
        out["triangle"][f"{measure}|{thres}"] = ref_loops.triangle_tsv(
            vcf, "6", tri_rows, names, measure, thres, ("EUR", "AMR"), ("male", "female"), ref_calc_ld)
queries = uniq[::3]
for ftype in ("tsv", "json", "rsids"):
    for measure, thres, flank in (("r_square", 0.8, 1000), ("d_prime", 0.9, 400), ("r_square", 0.05, 2500)):
        out["area"][f"{ftype}|{measure}|{thres}|{flank}"] = ref_loops.area_files(
            vcf, "6", queries, names, flank, measure, thres, ftype, ("ALL",), ("female",), ref_calc_ld)
# mixed ploidy: genotype lists of two lengths in one table (the reference zips them pair by pair)
rvcf, rnames = fakevcf.make_chromosome(haploid_from=24)
rrows = [[r.pos, r.id] for r in rvcf.records if r.id.startswith("rs") and ";" not in r.id]
seen, runiq = set(), []
for r in rrows:
    if r[1] not in seen:
        seen.add(r[1])
        runiq.append(r)
out["triangle_ragged"] = {}
out["area_ragged"] = {}
for measure, thres in (("r_square", None), ("d_prime", 0.3)):
    out["triangle_ragged"][f"{measure}|{thres}"] = ref_loops.triangle_tsv(
        rvcf, "6", runiq[8:30:2], rnames, measure, thres, ("EUR",), ("male", "female"), ref_calc_ld)
for ftype in ("tsv", "json"):
    for measure, thres, flank in (("r_square", 0.05, 900), ("d_prime", 0.9, 2500)):
        out["area_ragged"][f"{ftype}|{measure}|{thres}|{flank}"] = ref_loops.area_files(
            rvcf, "6", runiq[::4], rnames, flank, measure, thres, ftype, ("ALL",), ("male", "female"), ref_calc_ld)
(OUT / "driver_text.json").write_text(json.dumps(out, indent=0, sort_keys=True))
print({k: {kk: (len(v) if isinstance(v, str) else len(v)) for kk, v in d.items()} for k, d in out.items()})
