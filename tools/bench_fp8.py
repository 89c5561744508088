"""fp8 (e4m3) operand GEMM vs the bf16 GEMM on the tower shapes that take their A operand from a LayerNorm (qkv, fc1):
same kernels, same LDS bytes per K tile, half the LDS reads per MFMA."""
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
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, M, N, K, act in (("siglip qkv", frames * 729, 3456, 1152, L.ACT_NONE),
                               ("siglip fc1", frames * 729, 4352, 1152, L.ACT_GELU_TANH),
                               ("dino qkv", frames * 730, 4608, 1536, L.ACT_NONE),
                               ("dino fc1", frames * 730, 8192, 1536, L.ACT_SWIGLU)):
        x = torch.randn(M, K, device="cuda", generator=g)
        w = torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)
        b = torch.randn(N, device="cuda", generator=g)
        x16, w16 = x.to(dt), w.to(dt)
        sa = x.abs().amax(1) / 448.0
        sw = (w.abs().max() / 448.0).item()
        x8, w8 = (x / sa[:, None]).to(torch.float8_e4m3fn), (w / sw).to(torch.float8_e4m3fn)
        stats = torch.stack([torch.zeros_like(sa), sa * sw], 1).contiguous()
        c1 = torch.zeros(N, device="cuda")
        out = torch.empty(M, N // 2 if act == L.ACT_SWIGLU else N, device="cuda", dtype=dt)
        res = []
        for dbg in (None, "1"):
            if dbg:
                L.load().tdc_gemm_set_debug(int(dbg))
            t16 = timeit(lambda: ops.gemm(x16, w16, b, act=act, out=out), iters=10)
            t8 = timeit(lambda: ops.gemm(x8, w8, b, act=act, out=out, ln_stats=stats, ln_c1=c1, out_dtype=dt), iters=10)
            L.load().tdc_gemm_set_debug(0)
            res.append((t16, t8))
        t88 = None
        if name.endswith("fc1"):    # e4m3 output with analytic row scales (tdc_gemm_desc.out_fp8, fp8 level 3)
            out8 = torch.empty(M, out.shape[1], device="cuda", dtype=torch.uint8)
            st2 = torch.empty(M, 2, device="cuda")
            stn = torch.stack([x8.float().norm(dim=1), sa * sw], 1).contiguous()
            t88 = timeit(lambda: ops.gemm(x8, w8, b, act=act, out=out8, ln_stats=stn, ln_c1=c1, out_dtype=dt, out_stats=st2,
                                          out_w2max=float(w8.float().norm(dim=1).max()), out_bmax=float(b.abs().max()),
                                          out_wscale=1.0), iters=10)
        fl = 2.0 * M * N * K / 1e9
        print("%-11s bf16 %7.3f ms %7.1f TF/s | fp8 %7.3f ms %7.1f TF/s (%+5.1f %%) || no epilogue: bf16 %7.1f TF/s fp8 %7.1f TF/s"
              % (name, res[0][0], fl / res[0][0], res[0][1], fl / res[0][1], 100 * (res[0][0] / res[0][1] - 1),
                 fl / res[1][0], fl / res[1][1]), flush=True)
        if t88:
            print("            fp8 operands, e4m3 output: %7.3f ms %7.1f TF/s" % (t88, fl / t88), flush=True)
        del x, w, x16, w16, x8, w8, out


if __name__ == "__main__":
    main()
