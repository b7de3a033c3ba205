#!/bin/bash
# K-loop-only (LDX_ABLATE=5) timing of the tuning variants of the MFMA kernel
set -u
for v in "" _noaexp _nobexp _nobread _nobar _noexp _noall; do
  echo -n "variant=libldx$v "; LDX_LIB=$PWD/ld_tools_amd/libldx$v.so LDX_ABLATE=${AB:-5} timeout -k 10 200 python tools/gpu_tri.py 40000 5008 mfma 5 2>&1 | grep -v amdgpu.ids
done
