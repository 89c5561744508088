#!/bin/bash
# Same-box A/B of the residual-load ring depth of the 16-bit read-modify-write epilogue (TDC_RMW16_RING: 8 as shipped, 12, 16):
# the library rebuilt in place per value, tools/bench_gemm_epi.py on the eight tower GEMM types.  GPU box, repo root.
set -e
O=gpurun_out/r04n
mkdir -p $O
: > $O/rmw16_ring_ab.log
for r in ${RINGS:-8 12 16}; do
python - <<PY
import importlib.util
spec = importlib.util.spec_from_file_location("b", "tdc-video_amd/build.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.FILE_FLAGS["gemm.hip"] = b.FILE_FLAGS.get("gemm.hip", []) + ["-DTDC_RMW16_RING=$r"]
b.build(force=True, verbose=False)
PY
echo "== TDC_RMW16_RING = $r" >> $O/rmw16_ring_ab.log
timeout -k 10 300 python tools/bench_gemm_epi.py 512 2>&1 | grep -v 'WARNING\|amdgpu' >> $O/rmw16_ring_ab.log
done
cat $O/rmw16_ring_ab.log
