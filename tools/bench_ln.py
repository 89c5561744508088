"""LayerNorm (fp32 residual stream -> 16-bit GEMM operand) at the bench's sizes: GB/s of the 6 algorithmic bytes per element.
The buffers are far larger than the 256 MB Infinity Cache, and a GEMM-sized dummy write between runs evicts what is left."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import ops  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dtype = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    for rows, cols in ((frames * 729, 1152), (frames * 730, 1536)):
        x = torch.randn(rows, cols, device="cuda", generator=g)
        gm = torch.ones(cols, device="cuda"); bt = torch.zeros(cols, device="cuda")
        y = torch.empty(rows, cols, device="cuda", dtype=dtype)
        ms = timeit(lambda: ops.layernorm(x, gm, bt, 1e-6, cols, dtype, y16=y), iters=20)
        print("ln rows=%d cols=%d %8.3f ms  %7.1f GB/s" % (rows, cols, ms, rows * cols * 6 / ms / 1e6), flush=True)
        # as in the tower: behind the fp32 read-modify-write GEMM that produced x
        a = torch.randn(rows, cols, device="cuda", generator=g).to(dtype)
        w = (torch.randn(cols, cols, device="cuda", generator=g) / cols ** 0.5).to(dtype)
        bias = torch.zeros(cols, device="cuda")
        tot = 0.0
        for it in range(6):
            ops.gemm(a, w, bias=bias, res=x, out=x, out_f32=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.layernorm(x, gm, bt, 1e-6, cols, dtype, y16=y)
            e1.record()
            torch.cuda.synchronize()
            if it:
                tot += e0.elapsed_time(e1)
        ms = tot / 5
        print("   behind the residual GEMM: %8.3f ms  %7.1f GB/s" % (ms, rows * cols * 6 / ms / 1e6), flush=True)
        del a, w
        del x, y
        # the towers' 16-bit residual stream (round 4): fp16 rows in, bf16 rows out, 4 algorithmic bytes per element
        x = torch.randn(rows, cols, device="cuda", generator=g).half()
        y = torch.empty(rows, cols, device="cuda", dtype=dtype)
        ms = timeit(lambda: ops.layernorm(x, gm, bt, 1e-6, cols, dtype, y16=y), iters=20)
        print("   fp16 rows -> bf16 rows: %8.3f ms  %7.1f GB/s" % (ms, rows * cols * 4 / ms / 1e6), flush=True)
        del x, y


if __name__ == "__main__":
    main()
