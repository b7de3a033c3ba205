#!/usr/bin/env python3
"""Where the cycles of the dominant kernel go, from the PMC passes of tools/gpu_prof.sh.

    python tools/pmc_table.py gpurun_out/prof_<tag> [kernel-name-substring]

Reads the sq / sq2 / sq3 / sq4 passes (per-dispatch means of the most-launched kernel whose name contains the substring,
default "triangle_mfma_kernel") and prints, per SIMD and launch:

  T                       = GRBM_GUI_ACTIVE / 8          (the counter is summed over the 8 XCDs)
  matrix pipe busy        = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs      (cycles; 32 per FP4 MFMA)
  vector instructions     = SQ_ACTIVE_INST_VALU x 4 / 1024             (the SQ counts in quad-cycles; MFMA issue included)
  both at once            = SQ_VALU_MFMA_COEXEC_CYCLES / 1024
  => matrix only, vector only, neither;   CU has no wave at all = T - SQ_BUSY_CU_CYCLES / 256 CUs
and per WAVE (SQ_WAVE_CYCLES = issuing + issue-stalled + parked, quad-cycles): the share of the 2048 wave slots that is
occupied, and what an occupied slot does (by instruction class: vector, scalar, LDS, other; stalled at issue; parked in
s_waitcnt / s_barrier), plus the instruction mix per pair.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "triangle_mfma_kernel"
SIMDS, CUS, XCDS, SLOTS = 1024, 256, 8, 2048

by = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_sq*", "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            if needle in r["Kernel_Name"]:
                by[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
if not by:
    sys.exit(f"no counters for a kernel matching {needle!r} under {root}")
name, cs = max(by.items(), key=lambda kv: max(len(v) for v in kv[1].values()))
c = {k: sum(v) / len(v) for k, v in cs.items()}
n = {k: len(v) for k, v in cs.items()}
print(f"kernel: {name[:100]}")
print(f"dispatches per counter: min {min(n.values())}, max {max(n.values())}")
T = c["GRBM_GUI_ACTIVE"] / XCDS
mfma = c["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS
valu_all = c["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS
mfma_issue = c.get("SQ_INSTS_MFMA", 0.0) * 4 / SIMDS          # one quad-cycle of ACTIVE_INST_VALU per MFMA
valu = valu_all - mfma_issue
co = c.get("SQ_VALU_MFMA_COEXEC_CYCLES", float("nan")) / SIMDS
union = mfma + valu - co
cu_idle = T - c.get("SQ_BUSY_CU_CYCLES", float("nan")) / CUS
pct = lambda x: f"{x:12.0f} cycles  {100 * x / T:5.1f} %"   # noqa: E731
print(f"\nper SIMD and launch (T = GRBM_GUI_ACTIVE / 8 = {T:.0f} cycles)")
print(f"  matrix pipe busy, vector instruction executing too   {pct(co)}")
print(f"  matrix pipe busy alone                               {pct(mfma - co)}")
print(f"  vector instruction executing, matrix pipe idle       {pct(valu - co)}")
print(f"  neither pipe executing                               {pct(T - union)}")
print(f"      of which: the CU holds no wave (ramp and tail)    {pct(cu_idle)}")
print(f"      of which: waves resident, neither pipe           {pct(T - union - cu_idle)}")
print(f"  (sum = T by construction; matrix pipe busy {100 * mfma / T:.1f} %, vector instructions {100 * valu / T:.1f} % "
      f"+ {100 * mfma_issue / T:.1f} % MFMA issue slots)")
wave = c["SQ_WAVE_CYCLES"]
slots = SLOTS * T / 4
print(f"\nper wave slot (2 per SIMD; quad-cycles): occupied {100 * wave / slots:.1f} % of {slots:.3g}")
parts = [("issuing a vector instruction (MFMA included)", c.get("SQ_ACTIVE_INST_VALU")),
         ("issuing a scalar instruction", c.get("SQ_ACTIVE_INST_SCA")),
         ("issuing an LDS instruction", c.get("SQ_ACTIVE_INST_LDS")),
         ("issuing another instruction (branch, message, ...)", c.get("SQ_ACTIVE_INST_MISC")),
         ("stalled at issue (pipe busy, dependency: SQ_WAIT_INST_ANY)", c.get("SQ_WAIT_INST_ANY")),
         ("   of that on the LDS queue (SQ_WAIT_INST_LDS)", c.get("SQ_WAIT_INST_LDS")),
         ("parked in s_waitcnt / s_barrier (SQ_WAIT_ANY)", c.get("SQ_WAIT_ANY"))]
for label, v in parts:
    if v is not None:
        print(f"  {label:62s} {100 * v / wave:5.1f} % of the occupied time")
acc = sum(v for (l, v) in parts if v is not None and not l.startswith("   "))
print(f"  (these sum to {100 * acc / wave:.1f} %; SQ_ACTIVE_INST_ANY = {100 * c.get('SQ_ACTIVE_INST_ANY', float('nan')) / wave:.1f} %)")
pairs = None
cmd = os.path.join(root, "command.txt")
if os.path.exists(cmd):
    a = open(cmd).read().split()
    snps = int(a[a.index("--snps") + 1]) if "--snps" in a else (10000 if "bench.py" in a[0] else None)
    if snps:
        pairs = snps * (snps - 1) / 2
if pairs:
    print(f"\ninstructions per launch, as wave-instructions x 64 lanes per pair ({pairs:.4g} pairs)")
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM",
              "SQ_INSTS_BRANCH"):
        if k in c:
            print(f"  {k:20s} {c[k]:12.4g}   {c[k] * 64 / pairs:7.2f} per pair")
    if "SQ_INSTS_VALU_MFMA_MOPS_F6F4" in c:
        print(f"  SQ_INSTS_VALU_MFMA_MOPS_F6F4 {c['SQ_INSTS_VALU_MFMA_MOPS_F6F4']:.4g}")
with open(os.path.join(root, "pmc_table.json"), "w") as fh:
    json.dump({"kernel": name, "T_cycles": T, "counters_mean_per_dispatch": c}, fh, indent=1)
