#!/bin/bash
# drain stagger of the persistent GEMM (TDC_GEMM_STAGGER_NS): kernel time and epilogue stamps per start offset
set -e
O=gpurun_out
mkdir -p $O
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTDC_GEMM_DIAG -Wno-unused-result -o /tmp/gemm_stamps tools/gemm_stamps.cpp
export TDC_GEMM_PERSIST=1
{
for shape in "186624 3456 1152 0 0 0" "186624 1152 1152 0 1 1" "186880 8192 1536 3 0 0" "186880 1536 4096 0 1 1"; do
  for ns in 0 600 1200 1800 3000 5000; do
    echo "== stagger=$ns"
    TDC_GEMM_STAGGER_NS=$ns /tmp/gemm_stamps $shape 20 2>&1 | grep -v WARNING | head -3
  done
done
} > $O/gemm_stagger.log 2>&1
grep -E "stagger|TF/s|per WAVE" $O/gemm_stagger.log
