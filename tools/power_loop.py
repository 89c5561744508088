"""Runs ONE kernel type of the towers back to back for N seconds (python tools/power_loop.py gemm|attn64|attn72|ln|ln_gemm 6) so that
rocm-smi, sampled beside it (tools/power_by_kernel.sh), shows the clock and socket power that kernel type holds by itself - the
bench interleaves them every few milliseconds and a one-second power sample cannot tell them apart.  Prints the rate reached."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import ops  # noqa: E402

kind, secs = sys.argv[1], float(sys.argv[2])
g = torch.Generator(device="cuda").manual_seed(0)
B = 512
if kind == "gemm":          # DINOv2 qkv: plain 16-bit tiles
    M, N, K = B * 730, 4608, 1536
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fn = lambda: ops.gemm(a, w, bias=bias, out=out)
    work, unit = 2.0 * M * N * K / 1e12, "TFLOP/s"
elif kind in ("attn64", "attn72"):
    H, d, S = (24, 64, 730) if kind == "attn64" else (16, 72, 729)
    D = H * d
    ld = ops.pad64(3 * D)
    qkv = torch.randn(B * S, ld, device="cuda", generator=g).bfloat16()
    out = torch.empty(B * S, ops.pad64(D), device="cuda", dtype=torch.bfloat16)
    fn = lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], out, B, H, d, S, S, 1 / math.sqrt(d), S * ld, S * ld,
                               S * ld, S * out.stride(0))
    work, unit = 4.0 * B * H * S * S * d / 1e12, "TFLOP/s"
else:                        # the towers' LayerNorm: fp16 rows in, bf16 rows out (DINOv2 width)
    rows, cols = B * 730, 1536
    x = torch.randn(rows, cols, device="cuda", generator=g).half()
    y = torch.empty(rows, cols, device="cuda", dtype=torch.bfloat16)
    gam = torch.ones(cols, device="cuda"); bet = torch.zeros(cols, device="cuda")
    fn = lambda: ops.layernorm(x, gam, bet, 1e-6, cols, torch.bfloat16, y16=y, x16_kernel=True)
    work, unit = rows * cols * 4 / 1e12, "TB/s"
for _ in range(3):
    fn()
torch.cuda.synchronize()
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    n += 50
dt = time.time() - t0
print("%s: %d launches in %.2f s, %.3f ms each, %.1f %s" % (kind, n, dt, dt / n * 1e3, work * n / dt, unit), flush=True)
