#!/bin/bash
# fabric traffic (FETCH_SIZE x 2 + WRITE_SIZE) of the bench's GEMM launch list per tile-order group height TDC_GEMM_GROUP_M
set -e
SHAPES=${1:-tools/gemm_shapes_T512.txt}
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_gm
/opt/rocm/bin/hipcc -O2 -o gpurun_out/pmc_gm/gemm_pmc tools/gemm_pmc.cpp -Ltdc-video_amd -ltdc_hip -Wl,-rpath,$PWD/tdc-video_amd
for gm in 8 4 2; do
  OUT=gpurun_out/pmc_gm/gm$gm
  mkdir -p $OUT
  for c in FETCH_SIZE WRITE_SIZE; do
    TDC_GEMM_GROUP_M=$gm rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -o pmc_$c -- gpurun_out/pmc_gm/gemm_pmc $SHAPES 1 > $OUT/pmc_$c.log 2>&1 || tail -3 $OUT/pmc_$c.log
  done
  echo "== TDC_GEMM_GROUP_M=$gm"
  python tools/pmc_summary.py $OUT $SHAPES 2>&1 | head -8
done
