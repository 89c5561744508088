#!/bin/bash
# clock / power the chip holds while the bench runs: rocm-smi sampled once a second beside `bench.py` (one GPU)
O=gpurun_out
mkdir -p $O
python bench.py --no-cpu-baseline --steps 15 --warmup 2 > $O/clk_bench.json 2> $O/clk_bench.err &
BP=$!
: > $O/clk_samples.log
while kill -0 $BP 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*: //' | tr '\n' ' ' >> $O/clk_samples.log
  echo >> $O/clk_samples.log
  sleep 1
done
wait $BP
sort $O/clk_samples.log | uniq -c | sort -rn | head -30
tail -c 200 $O/clk_bench.json
