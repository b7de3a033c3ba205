#!/bin/bash
# in-kernel stamps of the ld_area band kernel (tuning build with -DLDX_TUNING -DLDX_STAMPS_ONLY): LIB=libldx_ts
set -u
F=$PWD/gpurun_out/st_area.bin
LDX_STAMPS=$F LDX_LIB=$PWD/ld_tools_amd/${LIB:-libldx_ts}.so timeout -k 10 300 python tools/gpu_exp.py area2 2>&1 | grep -v amdgpu.ids
python tools/stamps.py $F; rm -f $F
