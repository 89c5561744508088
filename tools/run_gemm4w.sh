#!/bin/bash
# 4-wave prototypes (LDS-DMA staging: gemm4w, register staging: gemm4r) against the library's 8-wave kernel on the same data
set -e
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gpurun_out/gemm4w tools/gemm4w_proto.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gpurun_out/gemm4r tools/gemm4r_proto.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTDC_GEMM_DIAG -Wno-unused-result -o gpurun_out/gemm_stamps tools/gemm_stamps.cpp
for shape in "186624 3584 1152" "186880 4608 1536" "186880 8192 1536"; do
  gpurun_out/gemm4w $shape 20 | grep -v spot
  gpurun_out/gemm4r $shape 20
  TDC_GEMM_DEBUG=1 gpurun_out/gemm_stamps $shape 0 0 0 20 | head -1
done
