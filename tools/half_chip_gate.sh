#!/bin/bash
# Gate of the "two half-chips" idea: do two processes, each confined to half of the CUs of
# every XCD (ROC_GLOBAL_CU_MASK) and each encoding half of the video, beat one process on the whole chip?
# Run from the repo root on the GPU box; writes gpurun_out/half_chip_gate.log.
set -o pipefail
O=gpurun_out/half_chip_gate.log
LO=0x$(printf 'f%.0s' $(seq 32))
HI=0x$(printf 'f%.0s' $(seq 32))$(printf '0%.0s' $(seq 32))
B="timeout -k 10 420 python bench.py --no-cpu-baseline --steps 6 --warmup 2"
pick() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']; print(sys.argv[2], 'frames', d['config']['frames'], 'fps', d['value'], 'ms', d['ms_per_step'], 'gemm_ms', r['gemm_ms_per_step'], 'gemm_TF', r['achieved'], 'attn_ms', r['attention']['ms_per_step'], 'ln_ms', r['layernorm_ms_per_step'])" "$1" "$2" >> $O; }
: > $O
echo "== 0. sanity: a masked process runs at all" >> $O
ROC_GLOBAL_CU_MASK=$LO timeout -k 5 120 python -c "import torch; x=torch.randn(1<<26,device='cuda'); torch.cuda.synchronize(); print('masked ok', float((x*2).sum()))" >> $O 2>&1 || { echo "masked process failed"; cat $O; exit 1; }
ROC_GLOBAL_CU_MASK=$HI timeout -k 5 120 python -c "import torch; x=torch.randn(1<<26,device='cuda'); torch.cuda.synchronize(); print('masked ok', float((x*2).sum()))" >> $O 2>&1 || { echo "masked process failed"; cat $O; exit 1; }
echo "== 1. whole chip, T=512" >> $O
$B --frames 512 > gpurun_out/hc_full.json 2> gpurun_out/hc_full.err && pick gpurun_out/hc_full.json full || exit 1
echo "== 2. whole chip, T=256 (tile-round quantisation of the half video alone)" >> $O
$B --frames 256 > gpurun_out/hc_full256.json 2> gpurun_out/hc_full256.err && pick gpurun_out/hc_full256.json full256 || exit 1
echo "== 3. ONE process on half of every XCD (mask $LO, grid 128), T=256, the other half idle" >> $O
ROC_GLOBAL_CU_MASK=$LO TDC_BENCH_PERSIST_GRID=128 $B --frames 256 > gpurun_out/hc_half_alone.json 2> gpurun_out/hc_half_alone.err && pick gpurun_out/hc_half_alone.json half_alone || exit 1
for rep in a b; do
echo "== 4$rep. TWO independent processes, disjoint halves, T=256 each, started 0.4 s apart" >> $O
ROC_GLOBAL_CU_MASK=$LO TDC_BENCH_PERSIST_GRID=128 $B --steps 10 --frames 256 > gpurun_out/hc_pair0$rep.json 2> gpurun_out/hc_pair0$rep.err &
P0=$!
sleep 0.4
ROC_GLOBAL_CU_MASK=$HI TDC_BENCH_PERSIST_GRID=128 $B --steps 10 --frames 256 > gpurun_out/hc_pair1$rep.json 2> gpurun_out/hc_pair1$rep.err &
P1=$!
wait $P0 || exit 1
wait $P1 || exit 1
pick gpurun_out/hc_pair0$rep.json pair0$rep; pick gpurun_out/hc_pair1$rep.json pair1$rep
done
echo "== 5. two independent processes WITHOUT masks, T=256 each (what the masks buy)" >> $O
$B --steps 10 --frames 256 > gpurun_out/hc_nomask0.json 2> gpurun_out/hc_nomask0.err &
P0=$!
sleep 0.4
$B --steps 10 --frames 256 > gpurun_out/hc_nomask1.json 2> gpurun_out/hc_nomask1.err &
P1=$!
wait $P0 || exit 1
wait $P1 || exit 1
pick gpurun_out/hc_nomask0.json nomask0; pick gpurun_out/hc_nomask1.json nomask1
echo "== 6. the sharded two-rank rehearsal on one GPU (gloo), masked halves, T=512" >> $O
TDC_BENCH_CU_SPLIT=1 TDC_BENCH_ONE_GPU=1 TDC_DIST_BACKEND=gloo $B --gpus 2 --frames 512 > gpurun_out/hc_rehearsal.json 2> gpurun_out/hc_rehearsal.err && pick gpurun_out/hc_rehearsal.json rehearsal_masked
echo "== 7. the same without masks" >> $O
TDC_BENCH_ONE_GPU=1 TDC_DIST_BACKEND=gloo $B --gpus 2 --frames 512 > gpurun_out/hc_rehearsal_nomask.json 2> gpurun_out/hc_rehearsal_nomask.err && pick gpurun_out/hc_rehearsal_nomask.json rehearsal_nomask
cat $O
