"""Vendor-library yardstick (not a product path): torch.matmul (hipBLASLt) on the tower GEMM shapes next to tdc_gemm."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tdc_video_amd
from tdc_video_amd import ops, weights

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

dev = torch.device("cuda", 0)
for (M, N, K) in [(373248, 1152, 1152), (373248, 3456, 1152), (373248, 4352, 1152), (373248, 1152, 4352),
                  (373760, 1536, 1536), (373760, 4608, 1536), (373760, 8192, 1536), (373760, 1536, 4096)]:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_lt = timeit(lambda: torch.matmul(A, W.t(), out=C))
    C2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_us = timeit(lambda: ops.gemm(A, W, out=C2))
    fl = 2.0 * M * N * K
    print("M=%d N=%d K=%d  hipblaslt %.3f ms %.0f TF | tdc %.3f ms %.0f TF" % (M, N, K, t_lt, fl / t_lt / 1e9, t_us, fl / t_us / 1e9), flush=True)
