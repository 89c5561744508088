#!/bin/bash
# Round-4 evidence on one MI355X box (run from the repo root through gpurun; TDC_COMMIT = the commit being measured):
# the default bench line (bf16 tower operands, fp16 residual stream, fp16 connector / Q-Former; incl. cpu_baseline), the same
# with the fp32 residual stream on the same box, the GEMM launch list of one step + per-shape times + its PMC counters (one
# counter per rocprofv3 pass, torch-free replay), the rocprofv3 kernel statistics of the bench command, the T = 64 shard and
# the audio configuration, the tower attention kernels side by side, a 2-rank run of the plain `python bench.py --gpus 2`.
set -e
export TMPDIR=/tmp
O=gpurun_out/r04
R=$PWD
mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
{
echo "== residual stream fp16 (default)"; timeout -k 10 300 python bench.py --no-cpu-baseline
echo "== residual stream fp32";           timeout -k 10 300 python bench.py --no-cpu-baseline --res fp32
} > $O/bench_res_ab.txt 2>> $O/bench_n1.err
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --dump-gemm-shapes $O/gemm_shapes_T512.txt --gemm-shape-times $O/gemm_shape_times.txt > $O/bench_dump.json 2>> $O/bench_n1.err
bash tools/run_gemm_pmc.sh $O/gemm_shapes_T512.txt > $O/gemm_pmc.log 2>&1 || tail -5 $O/gemm_pmc.log
cp gpurun_out/pmc/gemm_pmc_summary.json $O/gemm_pmc_summary.json || true
{
echo "== T=64 (per-rank shard of the 8-GPU job)";      timeout -k 10 300 python bench.py --frames 64 --steps 10 --warmup 3 --no-cpu-baseline
echo "== T=64, 336 px (BASELINE config 2)";            timeout -k 10 300 python bench.py --frames 64 --px 336 --steps 10 --warmup 3 --no-cpu-baseline
echo "== T=512 + audio (config 4)";                    timeout -k 10 300 python bench.py --audio --no-cpu-baseline
echo "== T=512, fp16 towers (the reference's inference type)"; timeout -k 10 300 python bench.py --dtype fp16 --no-cpu-baseline
} > $O/config_table.log 2>> $O/bench_n1.err
timeout -k 10 300 python tools/debug_attn_pw.py > $O/attention_pw_vs_default.log 2>&1
TDC_BENCH_ONE_GPU=1 TDC_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --frames 128 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_n2_self_launch_one_gpu.json 2> $O/bench_n2.err || tail -5 $O/bench_n2.err
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r04 -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/$O/bench_n1_under_rocprof.json 2> $R/$O/rocprof.err
cd $R
ls $O
