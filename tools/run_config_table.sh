#!/bin/bash
# The configuration table of DESIGN.md §5 on one box (bench lines only, no CPU baseline, no counter leg).  Every line is kept
# whole in gpurun_out/cfg/<name>.json; the summary shows frames/s, step time, the GEMM roofline fraction and the MFMA fraction of
# the Q-Former cross-attention block (the north_star's >= 40 % target is quoted at K = 144, T = 512; K = 16 is the reference's own
# default, context_token_num of the released checkpoints).
O=gpurun_out/cfg
mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline --pmc off "$@" > $O/$name.json 2> $O/$name.err || echo "$name failed"; python3 -c "
import json,sys
d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r['xattn_block']
print('%-28s %7.1f frames/s %8.1f ms  gemm frac %.4f  xattn_block %.3f ms frac %s  emitted %d' % ('$name', d['value'], d['ms_per_step'], r['frac'], x['ms_per_step'], x['frac'], d['config']['emitted_tokens']))"; }
run T512_K144 --steps 5
run T512_K144_via_mixin --steps 5 --via-mixin
run T512_K16 --K 16 --steps 5
run T512_K16_H3072 --K 16 --hidden 3072 --steps 5
run T512_K144_audio --audio
run T1024_H3072 --frames 1024 --hidden 3072
run T1024_H3072_fp8_l1 --frames 1024 --hidden 3072 --dtype fp8 --fp8-level 1
run T1024_H3072_fp8_l2 --frames 1024 --hidden 3072 --dtype fp8 --fp8-level 2
run T1024_H3072_fp8_l3 --frames 1024 --hidden 3072 --dtype fp8 --fp8-level 3
run T64_px336 --frames 64 --px 336 --steps 10 --warmup 3
run T64 --frames 64 --steps 10 --warmup 3
run T128 --frames 128 --steps 6 --warmup 2
run T256 --frames 256 --steps 4
run T512_fp16 --dtype fp16
run T512_dino_fp16 --dino-dtype fp16
run T512_res_fp32 --res fp32
