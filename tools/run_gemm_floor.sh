#!/bin/bash
# Main-loop floors of the persistent GEMM (diagnostics build, epilogue skipped): the full loop, the loop without its
# LDS-DMA staging (MFMAs on stale LDS data) and the loop without its MFMAs (staging + fragment reads + barriers only).
set -e
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTDC_GEMM_DIAG -Wno-unused-result -o gpurun_out/gemm_stamps tools/gemm_stamps.cpp
export TDC_GEMM_PERSIST=1 TDC_GEMM_DEBUG=1
for mode in 0 1 2 3; do
  echo "== TDC_GEMM_DIAGMODE=$mode (1: no staging, 2: no MFMAs, 3: neither)"
  export TDC_GEMM_DIAGMODE=$mode
  gpurun_out/gemm_stamps 186624 3456 1152 0 0 0 20 | head -2
  gpurun_out/gemm_stamps 186880 8192 1536 3 0 0 20 | head -2
done
