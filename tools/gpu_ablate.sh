#!/bin/bash
set -u
for a in 0 1 2 3 4 5 7; do echo "ablate=$a"; LDX_ABLATE=$a timeout -k 10 200 python tools/gpu_tri.py 40000 5008 mfma 3 2>&1 | grep -v amdgpu.ids; done
