#!/bin/bash
# Round 5, epilogue diet (bias as the accumulators' initial value, v_fma_mix for the fp16 residual): same-box A/B of the tree's
# library against tools/ab/libtdc_hip_base.so (round 4's GEMM) - per tower-GEMM type with / without epilogue, then the bench line.
set -e
rm -f gpurun_out/ab_base.log gpurun_out/ab_new.log
bash tools/lib_ab.sh timeout -k 10 300 python tools/bench_gemm_epi.py 512
mv gpurun_out/ab_base.log gpurun_out/r05_epi_ab_base.log; mv gpurun_out/ab_new.log gpurun_out/r05_epi_ab_new.log
bash tools/lib_ab.sh timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2
mv gpurun_out/ab_base.log gpurun_out/r05_bench_ab_base.log; mv gpurun_out/ab_new.log gpurun_out/r05_bench_ab_new.log
grep -h "towers\|==" gpurun_out/r05_epi_ab_base.log gpurun_out/r05_epi_ab_new.log
