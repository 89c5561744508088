#!/bin/bash
# Round-5 evidence on one MI355X box (run from the repo root through gpurun; TDC_COMMIT = the commit being measured):
# the default bench line (incl. cpu_baseline and the in-run counter leg), the same counters from the stand-alone replay
# (tools/run_gemm_pmc.sh) for comparison, per-shape GEMM times, rocprofv3 kernel statistics of the bench command, the T = 64
# shard / audio / fp16 configurations, the attention counters, the stage-error report, a 2-rank self-launch on the one GPU.
set -e
export TMPDIR=/tmp
O=gpurun_out/r05
R=$PWD
mkdir -p $O
timeout -k 10 600 python bench.py --pmc-out $O/gemm_pmc_inrun.json > $O/bench_n1.json 2> $O/bench_n1.err
timeout -k 10 300 python bench.py --no-cpu-baseline --pmc off --steps 1 --warmup 1 --dump-gemm-shapes $O/gemm_shapes_T512.txt --gemm-shape-times $O/gemm_shape_times.txt > $O/bench_dump.json 2>> $O/bench_n1.err
bash tools/run_gemm_pmc.sh $O/gemm_shapes_T512.txt > $O/gemm_pmc.log 2>&1 || tail -5 $O/gemm_pmc.log
cp gpurun_out/pmc/gemm_pmc_summary.json $O/gemm_pmc_summary.json || true
{
echo "== T=64 (per-rank shard of the 8-GPU job)";      timeout -k 10 300 python bench.py --frames 64 --steps 10 --warmup 3 --no-cpu-baseline --pmc off
echo "== T=64, 336 px (BASELINE config 2)";            timeout -k 10 300 python bench.py --frames 64 --px 336 --steps 10 --warmup 3 --no-cpu-baseline --pmc off
echo "== T=512 + audio (config 4)";                    timeout -k 10 300 python bench.py --audio --no-cpu-baseline --pmc off
echo "== T=512, fp16 towers (the reference's inference type)"; timeout -k 10 300 python bench.py --dtype fp16 --no-cpu-baseline --pmc off
echo "== T=512, fp32 residual stream";                 timeout -k 10 300 python bench.py --res fp32 --no-cpu-baseline --pmc off
} > $O/config_table.log 2>> $O/bench_n1.err
TDC_BENCH_ONE_GPU=1 TDC_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --frames 128 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_n2_self_launch_one_gpu.json 2> $O/bench_n2.err || tail -5 $O/bench_n2.err
timeout -k 10 600 python tools/parity_report.py > $O/parity_report.log 2>&1 || tail -5 $O/parity_report.log
bash tools/run_attn_pmc.sh > $O/attn_pmc.log 2>&1 || tail -5 $O/attn_pmc.log
cp gpurun_out/pmc_attn/summary.txt $O/attention_pmc_summary.txt || true
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r05 -- python3 $R/bench.py --no-cpu-baseline --pmc off --steps 2 --warmup 1 > $R/$O/bench_n1_under_rocprof.json 2> $R/$O/rocprof.err
cd $R
ls $O
