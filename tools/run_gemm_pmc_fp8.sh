#!/bin/bash
# MFMA-busy counters of the fp8-operand GEMM instances on the tower shapes (run on the MI355X box from the repo root)
set -e
export PMC_REPS=1
SHAPES=tools/gemm_shapes_T512_fp8.txt
OUT=gpurun_out/pmc_fp8
mkdir -p $OUT
/opt/rocm/bin/hipcc -O2 -o $OUT/gemm_pmc tools/gemm_pmc.cpp -Ltdc-video_amd -ltdc_hip -Wl,-rpath,$PWD/tdc-video_amd
export TMPDIR=/tmp
export TDC_PMC_FP8=1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -o pmc_$c -- $OUT/gemm_pmc $SHAPES 1 > $OUT/pmc_$c.log 2>&1 || tail -5 $OUT/pmc_$c.log
done
python tools/pmc_summary.py $OUT $SHAPES
