"""csrc/attention_pw.hip against the library's default tower kernel: agreement and time at the tower shape and at a long
sequence (where the per-item prologue is amortised: the main loop's own rate).  python tools/debug_attn_pw.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import ops  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402


def run(B, H, S, qkv, form, d=64):
    D = H * d
    ld = qkv.shape[1]
    out = torch.zeros(B * S, ops.pad64(D), device="cuda", dtype=qkv.dtype)
    fn = lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], out, B, H, d, S, S, 0.125, S * ld, S * ld, S * ld,
                               S * out.stride(0), form=form)
    fn()
    torch.cuda.synchronize()
    return out, fn


g = torch.Generator(device="cuda").manual_seed(0)
for (B, H, S) in [(8, 24, 4096), (512, 24, 730)]:
    D = H * 64
    qkv = torch.randn(B * S, ops.pad64(3 * D), device="cuda", generator=g).half()
    ref = None
    for form in (0, 2):
        out, fn = run(B, H, S, qkv, form)
        if ref is None:
            ref = out.float()
        ms = timeit(fn)
        print("B=%d H=%d S=%d form %d (%s): %8.3f ms %7.1f TFLOP/s  max diff vs the default form %.3e" %
              (B, H, S, form, "pw" if form == 2 else "default", ms, 4.0 * B * H * S * S * 64 / ms / 1e9,
               (out.float() - ref).abs().max().item()), flush=True)
