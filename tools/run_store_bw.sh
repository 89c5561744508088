#!/bin/bash
# C-tile drain microbenchmark (tools/store_bw.cpp): does the per-CU store rate depend on how many CUs store at once, is the
# limit per XCD or chip-wide, and does staggering the tile-column groups of an XCD by a drain time remove the contention?
set -e
O=gpurun_out
mkdir -p $O
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/store_bw tools/store_bw.cpp
{
echo "== all CUs / fewer CUs storing at once (no main loop between drains)"
for G in 256 128 64 32 8; do /tmp/store_bw $G 8 0; done
echo "== one XCD alone (32 CUs)"
/tmp/store_bw 256 8 0 0 0 0 0
/tmp/store_bw 256 8 0 0 0 0 3
echo "== with a ~12 us MFMA 'main loop' between drains: synchronised vs tile-column groups staggered"
for st in 0 600 1200 1800 2500; do /tmp/store_bw 256 12 0 0 24 $st; done
echo "== same, plain stores"
for st in 0 1200; do /tmp/store_bw 256 12 1 0 24 $st; done
} > $O/store_bw.log 2>&1
cat $O/store_bw.log
