"""LayerNorm fusion A/B per piece (DINO width, bf16): residual-stream GEMM with / without the x16 + partials output,
consumer GEMM with / without the fold, tdc_ln_finalize, and the LayerNorm kernel they replace."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import ops, lib as L  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    M, D = frames * 730, 1536
    x = torch.randn(M, D, device="cuda", generator=g)
    x16 = torch.empty(M, D, device="cuda", dtype=dt)
    part = torch.empty(D // 64, M, 2, device="cuda")
    stats = torch.empty(M, 2, device="cuda")
    gm = torch.ones(D, device="cuda"); bt = torch.zeros(D, device="cuda")
    for name, K in (("out (K=1536)", 1536), ("fc2 (K=4096)", 4096)):
        a = torch.randn(M, K, device="cuda", generator=g).to(dt)
        w = (torch.randn(D, K, device="cuda", generator=g) / math.sqrt(K)).to(dt)
        b = torch.randn(D, device="cuda", generator=g)
        t0 = timeit(lambda: ops.gemm(a, w, b, res=x, out=x, out_f32=True), iters=10)
        t1 = timeit(lambda: ops.gemm(a, w, b, res=x, out=x, out_f32=True, x16=x16, ln_part=part), iters=10)
        print("producer %-13s plain %7.3f ms | + x16 + partials %7.3f ms (%+.3f)" % (name, t0, t1, t1 - t0), flush=True)
        del a, w
    tf = timeit(lambda: ops.ln_finalize(part, D // 64, M, 1e-6, stats), iters=10)
    tl = timeit(lambda: ops.layernorm(x, gm, bt, 1e-6, D, dt, y16=x16), iters=10)
    print("tdc_ln_finalize %7.3f ms | LayerNorm kernel %7.3f ms" % (tf, tl), flush=True)
    ops.ln_finalize(part, D // 64, M, 1e-6, stats)
    for name, N, act in (("qkv (N=4608)", 4608, L.ACT_NONE), ("fc1 (N=8192 SwiGLU)", 8192, L.ACT_SWIGLU)):
        w = (torch.randn(N, D, device="cuda", generator=g) / math.sqrt(D)).to(dt)
        b = torch.randn(N, device="cuda", generator=g)
        c1 = w.float().sum(1).contiguous()
        out = torch.empty(M, N // 2 if act == L.ACT_SWIGLU else N, device="cuda", dtype=dt)
        t0 = timeit(lambda: ops.gemm(x16, w, b, act=act, out=out), iters=10)
        t1 = timeit(lambda: ops.gemm(x16, w, b, act=act, out=out, ln_stats=stats, ln_c1=c1), iters=10)
        print("consumer %-20s plain %7.3f ms | folded %7.3f ms (%+.3f)" % (name, t0, t1, t1 - t0), flush=True)
        del w, out


if __name__ == "__main__":
    main()
