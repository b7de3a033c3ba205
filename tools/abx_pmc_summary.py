#!/usr/bin/env python3
"""Per library build and shape: mean counters per launch of the dominant triangle / area kernel (tools/gpu_abx.sh PMC=1) and
the vector lane-instructions per pair that follow from them."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "*"))):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    tri = {k: v for k, v in acc.items() if "triangle" in k}
    if not tri:
        continue
    name, cs = max(tri.items(), key=lambda kv: max(len(x) for x in kv[1].values()))
    m = re.search(r"_(\d+)_(\d+)_", os.path.basename(d) + "_")
    line = os.path.basename(d) + ": " + " ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(cs.items()))
    if m and "SQ_INSTS_VALU" in cs:
        n = int(m.group(1))
        pairs = n * (n - 1) / 2
        valu = sum(cs["SQ_INSTS_VALU"]) / len(cs["SQ_INSTS_VALU"])
        line += f" | lane-instructions per pair = {valu * 64 / pairs:.2f}"
    print(line)
