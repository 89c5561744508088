"""Per tower-GEMM type (real epilogue: residual read-modify-write - fp16 stream by default, third argument "fp32" for the
fp32 stream of rounds 1-3 - / GELU / SwiGLU / plain 16-bit): time of the whole kernel vs the
same launch with the epilogue skipped (tdc_gemm_set_debug(1)), i.e. the share of the C-tile drain.  bf16, one MI355X.
With a second argument N: also the same launch under tdc_gemm_set_debug(N) (A/B of an epilogue experiment switch)."""
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
    alt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    res32 = len(sys.argv) > 3 and sys.argv[3] == "fp32"
    dtype = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    Ms, Md = frames * 729, frames * 730
    cases = [  # name, M, N, K, act, residual
        ("siglip qkv", Ms, 3456, 1152, L.ACT_NONE, False),
        ("siglip out", Ms, 1152, 1152, L.ACT_NONE, True),
        ("siglip fc1", Ms, 4352, 1152, L.ACT_GELU_TANH, False),
        ("siglip fc2", Ms, 1152, 4352, L.ACT_NONE, True),
        ("dino qkv", Md, 4608, 1536, L.ACT_NONE, False),
        ("dino out", Md, 1536, 1536, L.ACT_NONE, True),
        ("dino fc1", Md, 8192, 1536, L.ACT_SWIGLU, False),
        ("dino fc2", Md, 1536, 4096, L.ACT_NONE, True),
    ]
    tot = [0.0, 0.0]
    for name, M, N, K, act, res in cases:
        a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
        w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(dtype)
        bias = torch.randn(N, device="cuda", generator=g)
        n_out = N // 2 if act == L.ACT_SWIGLU else N
        if res:
            x = torch.randn(M, N, device="cuda", generator=g)
            if res32:
                fn = lambda: ops.gemm(a, w, bias=bias, res=x, out=x, out_f32=True)
            else:
                x = x.half()
                fn = lambda: ops.gemm(a, w, bias=bias, res=x, out=x)
        else:
            out = torch.empty(M, n_out, device="cuda", dtype=dtype)
            fn = lambda: ops.gemm(a, w, bias=bias, act=act, out=out)
        L.load().tdc_gemm_set_debug(0)
        timeit(fn, iters=10)          # the first timed batch on fresh buffers runs 10-13 % slow (first touch): discard it
        ms = ms0 = msa = 1e9
        for _ in range(2):            # alternate the forms, keep the better of two batches each
            L.load().tdc_gemm_set_debug(0)
            ms = min(ms, timeit(fn, iters=10))
            L.load().tdc_gemm_set_debug(1)
            ms0 = min(ms0, timeit(fn, iters=10))
            if alt:
                L.load().tdc_gemm_set_debug(alt)
                msa = min(msa, timeit(fn, iters=10))
        L.load().tdc_gemm_set_debug(0)
        fl = 2.0 * M * N * K
        layers = 27 if name.startswith("siglip") else 40
        tot[0] += ms * layers; tot[1] += ms0 * layers
        print("%-11s M=%6d N=%5d K=%5d  %7.3f ms %7.1f TF/s | no-epilogue %7.3f ms %7.1f TF/s | epilogue %4.1f %%%s"
              % (name, M, N, K, ms, fl / ms / 1e9, ms0, fl / ms0 / 1e9, 100 * (ms - ms0) / ms,
                 " | debug %d: %7.3f ms" % (alt, msa) if alt else ""), flush=True)
        del a, w
    print("towers (x layers): %.1f ms, without epilogues %.1f ms" % (tot[0], tot[1]))


if __name__ == "__main__":
    main()
