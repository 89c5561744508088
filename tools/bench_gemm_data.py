"""Does the GEMM rate depend on the operand VALUES (a power-limited chip clocks by what the multipliers toggle)?  One tower shape,
bf16, the same launch on differently distributed A operands."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import ops  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402


def main():
    M, N, K = 186880, 4608, 1536
    g = torch.Generator(device="cuda").manual_seed(0)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    base = torch.randn(M, K, device="cuda", generator=g)
    col = torch.exp(1.5 * torch.randn(K, device="cuda", generator=g))
    cases = [("N(0,1)", base), ("N(0,1) x 64", base * 64), ("N(0,1) + 30 (a large common mean)", base + 30.0),
             ("per-channel scales exp(1.5 N(0,1))", base * col), ("N(0,1)^3 (heavy tails)", base ** 3),
             ("rounded to 3 mantissa bits", (base * 4).round() / 4), ("zeros", base * 0)]
    for name, a32 in cases:
        a = a32.bfloat16()
        fn = lambda: ops.gemm(a, w, bias=bias, out=out)
        timeit(fn, iters=10)
        ms = min(timeit(fn, iters=10) for _ in range(2))
        print("%-40s %7.3f ms  %7.1f TFLOP/s" % (name, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
        del a


if __name__ == "__main__":
    main()
