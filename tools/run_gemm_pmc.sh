#!/bin/bash
# HBM-traffic and MFMA-busy counters (one counter per pass) of the GEMM kernels on the bench's shapes (run on the MI355X box from the repo root).
#   1. python bench.py --steps 1 --warmup 0 --no-cpu-baseline --dump-gemm-shapes gpurun_out/gemm_shapes.txt
#   2. bash tools/run_gemm_pmc.sh gpurun_out/gemm_shapes.txt
set -e
export PMC_REPS=${PMC_REPS:-2}     # launches per shape; the summary takes the last one
SHAPES=${1:-tools/gemm_shapes_T512.txt}
OUT=gpurun_out/pmc
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $OUT/gemm_pmc tools/gemm_pmc.cpp -Ltdc-video_amd -ltdc_hip -Wl,-rpath,$PWD/tdc-video_amd
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -o pmc_$c -- $OUT/gemm_pmc $SHAPES ${PMC_REPS:-2} > $OUT/pmc_$c.log 2>&1 || tail -5 $OUT/pmc_$c.log
done
python tools/pmc_summary.py $OUT $SHAPES
