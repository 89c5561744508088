#!/bin/bash
# clock and socket power each tower kernel type holds BY ITSELF (rocm-smi once a second beside a 7-second loop of that kernel):
# which phases of the step run against the power limit and which do not.  Run from the repo root on the GPU box.
O=gpurun_out
mkdir -p $O
: > $O/power_by_kernel.log
for k in gemm attn64 attn72 ln; do
  timeout -k 10 120 python tools/power_loop.py $k 7 > $O/power_$k.out 2> $O/power_$k.err &
  BP=$!
  echo "== $k" >> $O/power_by_kernel.log
  while kill -0 $BP 2>/dev/null; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power \(W\)" | sed 's/.*: //' | tr '\n' ' ' >> $O/power_by_kernel.log
    echo >> $O/power_by_kernel.log
    sleep 1
  done
  wait $BP
  cat $O/power_$k.out >> $O/power_by_kernel.log
done
cat $O/power_by_kernel.log
