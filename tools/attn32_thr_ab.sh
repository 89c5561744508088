#!/bin/bash
# Same-box A/B of the deferred-rescale threshold of attention32.hip (ATTN32_THR; 0 = the running maximum follows every tile, as in
# rounds 1-3): the library as shipped, then rebuilt in place with -DATTN32_THR=0.0f.  GPU box, repo root.
set -e
O=gpurun_out/r04e
mkdir -p $O
echo "== ATTN32_THR = 8 (as shipped)" > $O/attn32_thr_ab.log
timeout -k 10 200 python tools/bench_attn.py 512 2>/dev/null | grep "32x32" >> $O/attn32_thr_ab.log
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k attention > $O/attn_tests.log 2>&1 || { tail -20 $O/attn_tests.log; exit 1; }
tail -1 $O/attn_tests.log >> $O/attn32_thr_ab.log
python - <<'PY'
import importlib.util, os
spec = importlib.util.spec_from_file_location("b", "tdc-video_amd/build.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.FILE_FLAGS["attention32.hip"] = b.FILE_FLAGS["attention32.hip"] + ["-DATTN32_THR=0.0f"]
b.build(force=True, verbose=False)
PY
echo "== ATTN32_THR = 0 (rounds 1-3 behaviour)" >> $O/attn32_thr_ab.log
timeout -k 10 200 python tools/bench_attn.py 512 2>/dev/null | grep "32x32" >> $O/attn32_thr_ab.log
cat $O/attn32_thr_ab.log
