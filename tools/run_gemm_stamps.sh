#!/bin/bash
# builds the diagnostics GEMM (in-kernel stamps) on the GPU box and prints the tile timeline of the tower shapes
set -e
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTDC_GEMM_DIAG -Wno-unused-result -o gpurun_out/gemm_stamps tools/gemm_stamps.cpp
for persist in 0 1; do
  echo "== TDC_GEMM_PERSIST=$persist"
  export TDC_GEMM_PERSIST=$persist
  gpurun_out/gemm_stamps 186624 3456 1152 0 0 0 20
  gpurun_out/gemm_stamps 186624 1152 1152 0 1 1 20
  gpurun_out/gemm_stamps 186880 8192 1536 3 0 0 20
  gpurun_out/gemm_stamps 186880 1536 4096 0 1 1 20
done
