"""tdc_sva_attention at the bench's size (T = 512 frames, 144 queries per frame, 2 towers x 2 x 2 keys, C = 1024, 16 heads): ms
per launch and the HBM rate on its algorithmic bytes (q 2 KB + 8 x (K 2 KB + V 2 KB) + out 2 KB per query).  python tools/bench_sva.py [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import ops  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
side, r, dim, heads = 12, 2, 1024, 16
n = side * r
g = torch.Generator(device="cuda").manual_seed(0)
for dtype in (torch.float16, torch.bfloat16):
    q = torch.randn(T * side * side, dim, device="cuda", generator=g).to(dtype)
    kv = [torch.randn(T * n * n, 2 * dim, device="cuda", generator=g).to(dtype) for _ in range(2)]
    mask = torch.ones(T * side * side, 8, device="cuda", dtype=torch.uint8)
    out = torch.empty(T * side * side, dim, device="cuda", dtype=dtype)
    fn = lambda: ops.sva_attention(q, kv, mask, T, side, r, dim, heads, out=out)
    ms = timeit(fn)
    by = T * side * side * (2 * dim * 2 + 8 * 2 * dim * 2)
    print("sva T=%d %s: %.3f ms, %.2f TB/s on %.2f GB" % (T, dtype, ms, by / ms / 1e9, by / 1e9), flush=True)
