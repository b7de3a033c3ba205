#!/bin/bash
# A/B of LDX_ABLATE settings on the MFMA kernel (bit 8 = no stagger): usage gpu_ab.sh "0 8" [snps] [haps]
set -u
N=${2:-40000}; H=${3:-5008}
for a in $1; do echo -n "ablate=$a "; LDX_ABLATE=$a timeout -k 10 200 python tools/gpu_tri.py $N $H mfma 5 2>&1 | grep -v amdgpu.ids; done
