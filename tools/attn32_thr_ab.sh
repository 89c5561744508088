#!/bin/bash
# Same-box A/B of the deferred-rescale threshold of attention32.hip (ATTN32_THR; 0 = the running maximum follows every tile, as in
# rounds 1-3): the library rebuilt in place per value (THRS, default "8 0").  GPU box, repo root.
set -e
O=gpurun_out/r04e
mkdir -p $O
: > $O/attn32_thr_ab.log
for t in ${THRS:-8 0}; do
python - <<PY
import importlib.util
spec = importlib.util.spec_from_file_location("b", "tdc-video_amd/build.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.FILE_FLAGS["attention32.hip"] = b.FILE_FLAGS["attention32.hip"] + ["-DATTN32_THR=$t.0f"]
b.build(force=True, verbose=False)
PY
echo "== ATTN32_THR = $t" >> $O/attn32_thr_ab.log
timeout -k 10 200 python tools/bench_attn.py 512 2>/dev/null | grep "32x32" >> $O/attn32_thr_ab.log
done
cat $O/attn32_thr_ab.log
