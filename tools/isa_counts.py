#!/usr/bin/env python3
"""Instruction counts of the shipped product kernel's two hot loops, from the code object itself (llvm-objdump):

    python tools/isa_counts.py [ld_tools_amd/libldx.so]

  * one K-block of the 64-row unit's K loop (32 FP4 MFMAs), by instruction class;
  * one step of the fp32 epilogue tier (8 pairs), by instruction class, with the vector instructions by opcode.
Kernel: triangle_mfma_kernel<false, false, false, true, ldx_k16> (no side outputs, FP4, 4-byte cells).  The K-block is taken
between the 33rd and the 65th MFMA of the kernel's larger MFMA cluster (the second block of the 64-row loop), the step between
two consecutive pairs of 16-byte non-temporal stores of the n > 4096 instantiation of the step loop (the window includes
the not-taken park path)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from collections import Counter

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ld_tools_amd", "libldx.so")
objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
d = tempfile.mkdtemp()
shutil.copy(lib, os.path.join(d, "libldx.so"))
subprocess.run([objdump, "--offloading", "libldx.so"], cwd=d, capture_output=True, check=True)
ins = []
for obj in sorted(os.listdir(d)):
    if not obj.endswith("gfx950"):
        continue
    text = subprocess.run([objdump, "-d", os.path.join(d, obj)], capture_output=True, text=True).stdout.split("\n")
    func = None
    for ln in text:
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
        if m:
            func = m.group(1)
        elif ln.startswith("\t") and func and "triangle_mfma_kernelILb0ELb0ELb0ELb1E7ldx_k16" in func:
            ins.append(ln.split("//")[0].strip())
if not ins:
    sys.exit("kernel not found in " + lib)


def classes(seg):
    c = Counter(x.split()[0] for x in seg)
    g = Counter()
    for k, v in c.items():
        if k.startswith("v_mfma"):
            g["MFMA"] += v
        elif k.startswith("v_"):
            g["vector (VALU)"] += v
        elif k in ("s_waitcnt", "s_nop"):
            g[k] += v
        elif k.startswith("s_"):
            g["scalar"] += v
        elif k.startswith("ds_"):
            g["LDS"] += v
        else:
            g["vector memory"] += v
    return c, g


mf = [i for i, x in enumerate(ins) if x.startswith("v_mfma")]
clusters = []
for i in mf:
    if clusters and i - clusters[-1][-1] < 300:
        clusters[-1].append(i)
    else:
        clusters.append([i])
big = max(clusters, key=len)                       # the 64-row unit's loop: 96 MFMAs (three K-blocks per trip)
seg = ins[big[32]:big[64]]
c, g = classes(seg)
print(f"K loop, one K-block of a 64-row unit ({len(seg)} instructions):", dict(g))
print("   vector by opcode:", sorted(((k, v) for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma")), key=lambda kv: -kv[1]))
st = [i for i, x in enumerate(ins) if x.startswith("global_store_dwordx4") and " nt" in x]
pairs = [(a, b) for a, b in zip(st, st[1:]) if b - a < 20]       # the two 16-byte stores of one step of a 64-row unit
for a, b in pairs:                                                # one pair per instantiation of the step loop (n <= 4096, n > 4096)
    rd = [i for i in range(max(0, a - 400), a) if ins[i].startswith("ds_read_b128") and any(ins[j].startswith("ds_read_b128") for j in (i + 1, i + 2))]
    if not rd:
        continue
    seg = ins[rd[-1]:b + 1]                                       # from the step's row-operand reads to its second store
    c, g = classes(seg)
    if g["vector (VALU)"] < 150 or any(k.endswith("_f64") or "_f64_" in k for k in c):
        continue                                                  # (a store pair of another code path)
    print(f"fp32 tier, one step of a 64-row unit = 8 pairs, from its row-operand reads to its second store ({len(seg)} instructions; the "
          f"park test and the loop control behind it are ~30 more, mostly scalar):", dict(g))
    print("   vector by opcode:", sorted(((k, v) for k, v in c.items() if k.startswith("v_")), key=lambda kv: -kv[1]))
