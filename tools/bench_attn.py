"""Attention micro-benchmark on the tower shapes (bf16): TFLOP/s on the real head dim + max error vs torch SDPA (fp32)."""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import ops  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402


def main():
    dtype = torch.bfloat16
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    g = torch.Generator(device="cuda").manual_seed(0)
    for (H, d, S) in [(16, 72, 729), (24, 64, 730)]:
        D = H * d
        ld = ops.pad64(3 * D)
        qkv = torch.randn(B * S, ld, device="cuda", generator=g).to(dtype)
        out = torch.empty(B * S, ops.pad64(D), device="cuda", dtype=dtype)
        fn = lambda form=0: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], out, B, H, d, S, S,
                                          1 / math.sqrt(d), S * ld, S * ld, S * ld, S * out.stride(0), form=form)
        ms16 = timeit(lambda: fn(1))
        ms = timeit(fn)
        q, k, v = (qkv[: 2 * S, i * D:(i + 1) * D].float().view(2, S, H, d).transpose(1, 2) for i in range(3))
        ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(2 * S, D)
        err = (out[: 2 * S, :D].float() - ref).abs().max().item()
        fl = 4.0 * B * H * S * S * d
        print("attn B=%d H=%d d=%d S=%d  32x32 form %8.3f ms  %7.1f TFLOP/s | 16x16 form %8.3f ms  %7.1f TFLOP/s | max err %.2e"
              % (B, H, d, S, ms, fl / ms / 1e9, ms16, fl / ms16 / 1e9, err),
              flush=True)


if __name__ == "__main__":
    main()
