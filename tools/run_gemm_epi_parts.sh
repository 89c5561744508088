#!/bin/bash
# Where does the C-tile epilogue of the persistent GEMM spend its time?  Diagnostics build (TDC_GEMM_DIAG), same launch with
#   debug 0 = normal | 1 = epilogue skipped | 8 = no global stores (VALU + LDS staging only) | 16 = no LDS staging (stores
#   of register data to the same addresses) | 24 = neither | 32 = no residual loads (fp32 read-modify-write tiles)
set -e
O=gpurun_out
mkdir -p $O
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTDC_GEMM_DIAG -Wno-unused-result -o /tmp/gemm_stamps tools/gemm_stamps.cpp
export TDC_GEMM_PERSIST=1
{
for shape in "186624 3456 1152 0 0 0" "186880 4608 1536 0 0 0" "186624 1152 1152 0 1 1" "186880 1536 4096 0 1 1"; do
  for dbg in 0 1 8 16 24 32 40; do
    echo "== debug=$dbg"
    TDC_GEMM_DEBUG=$dbg /tmp/gemm_stamps $shape 20 2>&1 | grep -v WARNING | head -3
  done
done
} > $O/gemm_epi_parts.log 2>&1
cat $O/gemm_epi_parts.log
