"""Development aid for csrc/attention_pw.hip: correctness against the 32x32 form and timings at the tower shape and at a long sequence
(where the per-item prologue is amortised: the main loop's own rate)."""
import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tdc_video_amd  # noqa
from tdc_video_amd import ops
from tools.bench_ops import timeit


def run(B, H, S, qkv, form, d=64):
    D = H * d
    ld = qkv.shape[1]
    out = torch.zeros(B * S, ops.pad64(D), device="cuda", dtype=qkv.dtype)
    fn = lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], out, B, H, d, S, S, 0.125, S * ld, S * ld, S * ld,
                               S * out.stride(0), form=form)
    fn()
    torch.cuda.synchronize()
    return out, fn


forms = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 66]
g = torch.Generator(device="cuda").manual_seed(0)
for (B, H, S) in [(8, 24, 4096), (512, 24, 730)]:
    D = H * 64
    qkv = torch.randn(B * S, ops.pad64(3 * D), device="cuda", generator=g).half()
    ref = None
    for form in forms:
        out, fn = run(B, H, S, qkv, form)
        if ref is None:
            ref = out.float()
        ms = timeit(fn)
        print("B=%d H=%d S=%d form %3d: %8.3f ms %7.1f TFLOP/s  max diff vs first form %.3e" %
              (B, H, S, form, ms, 4.0 * B * H * S * S * 64 / ms / 1e9, (out.float() - ref).abs().max().item()), flush=True)

# cycle stamps (timing builds): cycles per iteration in [sync, slots 0-7, 8-15, 16-23, 24-31, tail]
B, H, S = 8, 24, 4096
D = H * 64
qkv = torch.randn(B * S, ops.pad64(3 * D), device="cuda", generator=g).half()
for form in (2 + 0x2000, 2 + 0x2100, 2 + 0x3000):
    out, fn = run(B, H, S, qkv, form)
    st = out.view(-1)[:32].view(torch.int64)[:7].tolist()
    nt = max(1, st[6])
    print("stamps form %#x: per-iteration cycles sync %.0f | slots 0-7 %.0f | 8-15 %.0f | 16-23 %.0f | 24-31 %.0f | tail %.0f | sum %.0f"
          % (form - 2, *[x / nt for x in st[:6]], sum(st[:6]) / nt))
