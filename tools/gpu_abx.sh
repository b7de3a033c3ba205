#!/bin/bash
# Same-box A/B of library builds -- ONE script for what gpu_ab.sh, gpu_ab2.sh, gpu_abl.sh, gpu_ablate.sh, gpu_libs.sh,
# gpu_times.sh, gpu_quick.sh and gpu_area_ab.sh each did a part of (round 5).  Boxes of the pool differ
# by up to 8 % and so do the first and the tenth second on one box, so builds are compared INTERLEAVED and REPEATED in one
# visit.  Everything is an environment variable:
#   LIBS    library builds under ld_tools_amd/, without ".so"      (default "libldx_base libldx")
#   SHAPES  "snps haps path reps fmt|..." for tools/gpu_tri.py     (default: the three judged shapes, FP4, k16)
#   ROUNDS  repetitions of the whole sweep                         (default 3)
#   ABL     LDX_ABLATE values per library (tuning builds only)     (default "0")
#   CHECK   "" (none) | a pytest -k expression run once per library before the timings (e.g. "triangle or agree", "area")
#   AREA    1: also time configs[2] (tools/gpu_exp.py area2) per round and library
#   PMC     1: afterwards one rocprofv3 pass per library and shape with SQ_INSTS_VALU (+ the r05 instruction-mix counters):
#           lane-instructions per pair of each build, from the same visit
# A step that is KILLED (timeout) ends the visit.
set -u
mkdir -p gpurun_out
LIBS=${LIBS:-libldx_base libldx}
SHAPES=${SHAPES:-"10000 5008 fp4 200 k16|40000 5008 fp4 10 k16|50000 1008 fp4 10 k16"}
ROUNDS=${ROUNDS:-3}
export TMPDIR=/tmp
timeout -k 10 600 python -c "import torch; print('torch', torch.__version__, torch.cuda.is_available())" 2>&1 | tail -1
if [ -n "${CHECK:-}" ]; then
  for v in $LIBS; do
    LDX_LIB=$PWD/ld_tools_amd/$v.so timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "$CHECK" > gpurun_out/abx_pytest_$v.log 2>&1; rc=$?
    echo "[check $v] exit $rc: $(tail -1 gpurun_out/abx_pytest_$v.log)"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  done
fi
IFS='|' read -ra S <<< "$SHAPES"
for r in $(seq 1 $ROUNDS); do
  for shape in "${S[@]}"; do
    for v in $LIBS; do
      for a in ${ABL:-0}; do
        echo -n "round=$r lib=$v ablate=$a "
        LDX_ABLATE=$a LDX_LIB=$PWD/ld_tools_amd/$v.so timeout -k 10 300 python tools/gpu_tri.py $shape 2>&1 | grep -v amdgpu.ids; rc=${PIPESTATUS[0]}
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
      done
    done
  done
  if [ "${AREA:-0}" = 1 ]; then
    for v in $LIBS; do
      echo -n "round=$r lib=$v "
      LDX_LIB=$PWD/ld_tools_amd/$v.so timeout -k 10 300 python tools/gpu_exp.py area2 2>&1 | grep -v amdgpu.ids
    done
  fi
done
if [ "${PMC:-0}" = 1 ]; then
  for shape in "${S[@]}"; do
    tag=$(echo $shape | tr ' ' '_')
    for v in $LIBS; do
      for p in sq sq2; do
        if [ $p = sq ]; then C="SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE";
        else C="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; fi
        LDX_LIB=$PWD/ld_tools_amd/$v.so timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d gpurun_out/abx_pmc/${v}_${tag}_$p -- python3 tools/gpu_tri.py $shape > gpurun_out/abx_pmc_${v}_${tag}_$p.log 2>&1
        echo "[pmc $p $v $shape] exit $?"
      done
    done
  done
  python3 tools/abx_pmc_summary.py gpurun_out/abx_pmc
fi
