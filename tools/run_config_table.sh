#!/bin/bash
# The configuration table of DESIGN.md §5 on one box (bench lines only, no CPU baseline)
O=gpurun_out/cfg
mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || echo "$name failed"; python3 -c "
import json,sys
d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('%-28s %7.1f frames/s %8.1f ms' % ('$name', d['value'], d['ms_per_step']))"; }
run T512_K144
run T512_K144_audio --audio
run T1024_H3072 --frames 1024 --hidden 3072
run T1024_H3072_fp8_l3 --frames 1024 --hidden 3072 --dtype fp8 --fp8-level 3
run T64_px336 --frames 64 --px 336
run T64 --frames 64
run T128 --frames 128
run T256 --frames 256
run T512_K16 --K 16
run T512_fp16 --dtype fp16
