set -e
run() { name=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline --pmc off --steps 2 --warmup 1 "$@" > gpurun_out/combo_$name.json 2> gpurun_out/combo_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/combo_$name.err; }; python3 -c "
import json; d=json.loads(open('gpurun_out/combo_$name.json').read().strip().splitlines()[-1]); print('%-34s %7.1f frames/s  emitted %d  %s' % ('$name', d['value'], d['config']['emitted_tokens'], d['config']['entry'][:6]))"; }
run K16_mixin --K 16 --via-mixin
run K16_engine --K 16
run px336_T64_mixin --frames 64 --px 336 --via-mixin
run px336_T64_engine --frames 64 --px 336
run fp8l3_T1024_H3072_mixin --frames 1024 --hidden 3072 --dtype fp8 --fp8-level 3 --via-mixin
run fp8l3_T1024_H3072_engine --frames 1024 --hidden 3072 --dtype fp8 --fp8-level 3
run dino_fp16_mixin --dino-dtype fp16 --via-mixin
run fp16_fuse_mixin --dtype fp16 --ln-fuse --via-mixin
