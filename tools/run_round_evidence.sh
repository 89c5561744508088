set -e
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/h_all.log 2>&1
timeout -k 10 400 python bench.py > gpurun_out/h_bench.json 2> gpurun_out/h_bench.err
timeout -k 10 300 python bench.py --dtype fp8 --fp8-level 3 --no-cpu-baseline > gpurun_out/h_bench_fp8_l3.json 2> gpurun_out/h_bench_fp8.err
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/h_prof -o h -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/h_bench_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/h_rocprof.err
