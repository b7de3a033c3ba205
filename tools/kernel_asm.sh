#!/bin/bash
# Extract one kernel's ISA from the -save-temps output of `python ld_tools_amd/build.py --save-temps`.
# usage: tools/kernel_asm.sh <mangled-name-regex> [out.s]   (always run from the repo root)
set -eu
F=/root/repo/ld_tools_amd/build/ldx_mfma-hip-amdgcn-amd-amdhsa-gfx950.s
[ -n "${SRC:-}" ] && F=$SRC
L=$(grep -n "^$1.*:" "$F" | head -1 | cut -d: -f1)
[ -n "$L" ] || { echo "kernel not found: $1" >&2; exit 1; }
awk -v s="$L" 'NR>=s{print} NR>s && /s_endpgm/{exit}' "$F" > "${2:-/tmp/kernel.s}"
wc -l "${2:-/tmp/kernel.s}"
