"""How long does the GEMM store epilogue take per tile, as a function of how many CUs write at once?"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tdc_video_amd  # noqa
from tdc_video_amd import ops
from tools.bench_ops import timeit

dtype = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in [(2048, 2048, 1152), (4096, 4096, 1152), (8192, 8192, 1152), (16384, 16384, 1152), (16384, 16384, 128),
                  (2048, 2048, 128)]:
    a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(dtype)
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    ms = timeit(lambda: ops.gemm(a, w, out=out), iters=30)
    tiles = (M // 256) * (N // 256)
    print("M=%5d N=%5d K=%4d tiles=%5d rounds=%5.2f  %.4f ms  %.2f us per round" % (M, N, K, tiles, tiles / 256, ms, ms * 1e3 / math.ceil(tiles / 256)))
