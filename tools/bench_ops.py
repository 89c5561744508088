"""Micro-benchmarks of the hot kernels (GEMM, attention, LayerNorm) on one MI355X; prints TFLOP/s / GB/s."""
import math
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dtype = torch.bfloat16 if "--bf16" in sys.argv else torch.float16
    g = torch.Generator(device="cuda").manual_seed(0)
    print("dtype", dtype)
    for (M, N, K) in [(46656, 3456, 1152), (46656, 1152, 1152), (46656, 4352, 1152), (46656, 1152, 4352),
                      (46720, 4608, 1536), (46720, 1536, 1536), (46720, 8192, 1536), (46720, 1536, 4096),
                      (36864, 2048, 1024), (9216, 3584, 1024), (9216, 3584, 3584), (8192, 8192, 8192),
                      (68016, 9216, 3584), (4096, 4096, 4096), (46656, 4352, 128), (46656, 4352, 256),
                      (373248, 3456, 1152), (373248, 1152, 1152), (373248, 4352, 1152), (373248, 1152, 4352)]:
        a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
        w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(dtype)
        out = torch.empty(M, N, device="cuda", dtype=dtype)
        ms = timeit(lambda: ops.gemm(a, w, out=out))
        print("gemm M=%6d N=%5d K=%5d  %8.3f ms  %7.1f TFLOP/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
        del a, w, out
    for (B, H, d, S) in [(64, 16, 72, 729), (64, 24, 64, 730), (64, 16, 72, 576), (436, 12, 64, 28)]:
        D = H * d
        ld = ops.pad64(3 * D)
        qkv = torch.randn(B * S, ld, device="cuda", generator=g).to(dtype)
        out = torch.empty(B * S, ops.pad64(D), device="cuda", dtype=dtype)
        fn = lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], out, B, H, d, S, S,
                                   1 / math.sqrt(d), S * ld, S * ld, S * ld, S * out.stride(0))
        ms = timeit(fn)
        print("attn B=%d H=%d d=%d S=%d  %8.3f ms  %7.1f TFLOP/s" % (B, H, d, S, ms, 4.0 * B * H * S * S * d / ms / 1e9))
    for cols in (1152, 1536):
        rows = 46656
        x = torch.randn(rows, cols, device="cuda", generator=g)
        gm = torch.ones(cols, device="cuda"); bt = torch.zeros(cols, device="cuda")
        y = torch.empty(rows, cols, device="cuda", dtype=dtype)
        ms = timeit(lambda: ops.layernorm(x, gm, bt, 1e-6, cols, dtype, y16=y))
        print("ln rows=%d cols=%d %8.3f ms  %7.1f GB/s" % (rows, cols, ms, rows * cols * 6 / ms / 1e6))


if __name__ == "__main__":
    main()
