#!/bin/bash
# Round-3 evidence on one MI355X box (run from the repo root through gpurun; TDC_COMMIT = the commit being measured):
# the bench line in its default type (bf16 towers + fp16 connector / Q-Former, incl. cpu_baseline) and in plain fp16 / bf16, the
# same with the Q-Former cross-attention block as the per-kernel sequence, the GEMM launch list of one step + its PMC counters
# (one counter per rocprofv3 pass, torch-free replay), the rocprofv3 kernel statistics of the bench command, the other BASELINE
# configurations, micro-benchmarks of the cross-attention block and of the tower attention, a 2-rank rehearsal of the sharded
# bench on the one GPU (gloo).
set -e
export TMPDIR=/tmp
O=gpurun_out/r03
R=$PWD
mkdir -p $O
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
timeout -k 10 300 python bench.py --dtype fp16 --no-cpu-baseline > $O/bench_n1_fp16.json 2>> $O/bench_n1.err
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_n1_bf16.json 2>> $O/bench_n1.err
timeout -k 10 300 python bench.py --xattn-mode 0 --no-cpu-baseline > $O/bench_n1_xattn_mode0.json 2>> $O/bench_n1.err
timeout -k 10 300 python bench.py --xattn-mode 2 --no-cpu-baseline > $O/bench_n1_xattn_mode2.json 2>> $O/bench_n1.err
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --dump-gemm-shapes $O/gemm_shapes_T512.txt --gemm-shape-times $O/gemm_shape_times.txt > $O/bench_dump.json 2>> $O/bench_n1.err
bash tools/run_gemm_pmc.sh $O/gemm_shapes_T512.txt > $O/gemm_pmc.log 2>&1 || tail -5 $O/gemm_pmc.log
cp gpurun_out/pmc/gemm_pmc_summary.json $O/gemm_pmc_summary.json || true
{
echo "== T=64 (per-rank shard of the 8-GPU job)";      timeout -k 10 300 python bench.py --frames 64 --steps 10 --warmup 3 --no-cpu-baseline
echo "== T=64, 336 px (BASELINE config 2)";            timeout -k 10 300 python bench.py --frames 64 --px 336 --steps 10 --warmup 3 --no-cpu-baseline
echo "== T=512 + audio (config 4)";                    timeout -k 10 300 python bench.py --audio --no-cpu-baseline
echo "== T=1024, H=3072 (config 5 width)";             timeout -k 10 300 python bench.py --frames 1024 --hidden 3072 --steps 2 --no-cpu-baseline
for l in 1 3; do
echo "== fp8 level $l";                                 timeout -k 10 300 python bench.py --dtype fp8 --fp8-level $l --no-cpu-baseline
done
echo "== T=1024, H=3072, fp8 level 3 (config 5)";      timeout -k 10 300 python bench.py --frames 1024 --hidden 3072 --steps 2 --dtype fp8 --fp8-level 3 --no-cpu-baseline
} > $O/config_table.log 2>> $O/bench_n1.err
timeout -k 10 300 python tools/bench_xattn.py > $O/xattn_block_micro.log 2>&1
timeout -k 10 300 python tools/bench_xattn.py 439 144 156 bf16 >> $O/xattn_block_micro.log 2>&1
timeout -k 10 300 python tools/xattn_diag.py 0 1 2 4 7 15 > $O/xattn_kernel_parts.log 2>&1 || true
rm -f gpurun_out/xattn_diag_*.so
timeout -k 10 300 python tools/bench_attn.py 512 > $O/attention_micro.log 2>&1
# sharded bench rehearsal: 2 ranks on the one GPU, gloo transport (the driver runs the real N = 2/4/8 over RCCL)
TDC_BENCH_ONE_GPU=1 TDC_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --frames 128 --steps 3 --warmup 1 > $O/bench_n2_rehearsal.json 2> $O/bench_n2.err || tail -5 $O/bench_n2.err
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r03 -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/$O/bench_n1_under_rocprof.json 2> $R/$O/rocprof.err
cd $R
ls $O
