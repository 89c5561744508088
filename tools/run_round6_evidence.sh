#!/bin/bash
# Round-6 evidence on one MI355X box (run from the repo root through gpurun): the default bench line (incl. cpu_baseline and the
# in-run counter leg), the same step through the drop-in boundary (--via-mixin), per-shape GEMM times, rocprofv3 kernel statistics
# of the bench command, a 2-rank self-launch through the mixin on the one GPU (gloo).
set -e
export TMPDIR=/tmp
O=gpurun_out/r06
R=$PWD
mkdir -p $O
timeout -k 10 600 python bench.py --steps 10 --warmup 2 --pmc-out $O/gemm_pmc_inrun.json > $O/bench_n1.json 2> $O/bench_n1.err
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --pmc off --via-mixin > $O/bench_via_mixin.json 2>> $O/bench_n1.err
timeout -k 10 300 python bench.py --no-cpu-baseline --pmc off --steps 1 --warmup 1 --dump-gemm-shapes $O/gemm_shapes_T512.txt --gemm-shape-times $O/gemm_shape_times.txt > $O/bench_dump.json 2>> $O/bench_n1.err
TDC_BENCH_ONE_GPU=1 TDC_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --frames 128 --steps 3 --warmup 1 --no-cpu-baseline --via-mixin > $O/bench_n2_via_mixin_one_gpu.json 2> $O/bench_n2.err || tail -5 $O/bench_n2.err
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r06 -- python3 $R/bench.py --no-cpu-baseline --pmc off --steps 2 --warmup 1 > $R/$O/bench_n1_under_rocprof.json 2> $R/$O/rocprof.err
cd $R
ls $O
