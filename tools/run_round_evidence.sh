#!/bin/bash
# Round evidence on one MI355X box (run from the repo root through gpurun): the whole GPU test suite, the bench in bf16 and
# at the three fp8 levels on the same box, the fp8 GEMM micro-benchmark, and the rocprofv3 kernel statistics of the bench.
set -e
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/h_all.log 2>&1
timeout -k 10 400 python bench.py > $O/h_bench.json 2> $O/h_bench.err
for l in 1 2 3; do
  timeout -k 10 300 python bench.py --dtype fp8 --fp8-level $l --no-cpu-baseline > $O/h_bench_fp8_l$l.json 2> $O/h_bench_fp8.err
done
timeout -k 10 300 python tools/bench_fp8.py 256 > $O/h_fp8_gemm.log 2>&1
timeout -k 10 300 python tools/bench_attn.py > $O/h_attn.log 2>&1
R=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/h_prof -o h -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/$O/h_bench_rocprof.json 2> $R/$O/h_rocprof.err
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/h_prof8 -o h8 -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --dtype fp8 --fp8-level 3 > $R/$O/h_bench8_rocprof.json 2> $R/$O/h_rocprof8.err
