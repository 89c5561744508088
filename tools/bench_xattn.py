"""The Q-Former cross-attention BLOCK at the bench's size, fused against un-fused, bf16 / fp16, one MI355X:
  fused    key GEMM [F*N, 4608] + transposed value GEMM [4608, F*N] + 6 x tdc_qformer_xattn
  un-fused stacked K/V GEMM [F*N, 9216] + 6 x {q GEMM, tdc_attention, dense GEMM (fp32 residual), tdc_layernorm}
  out-fused (the default, xattn_mode 1) stacked K/V GEMM + 6 x {q GEMM, tdc_attention, tdc_qformer_xattn in its ctx form}
Algorithmic work: 12.76 GFLOP per compressed frame at K = 144, N = 156, H = 3584.
usage: python tools/bench_xattn.py [F=439] [K=144] [N=156] [dtype=fp16]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import ops  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 439
K = int(sys.argv[2]) if len(sys.argv) > 2 else 144
N = int(sys.argv[3]) if len(sys.argv) > 3 else 156
dt = torch.bfloat16 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else torch.float16
D, heads, H, Lt, NL = 768, 12, 3584, 12, 6
S = K + Lt
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*s, sc=1.0):
    return torch.randn(*s, device=dev, generator=g) * sc


enc = rnd(F * N, H).to(dt)
wk, wv = rnd(NL * D, H, sc=0.02).to(dt), rnd(NL * D, H, sc=0.02).to(dt)
wkv = torch.cat([torch.cat([wk[j * D:(j + 1) * D], wv[j * D:(j + 1) * D]]) for j in range(NL)]).contiguous()
bk, bkv, bv = rnd(NL * D, sc=0.02), rnd(2 * NL * D, sc=0.02), rnd(NL * D, sc=0.02)
wq = [rnd(D, D, sc=0.03).to(dt) for _ in range(NL)]
wo = [rnd(D, D, sc=0.03).to(dt) for _ in range(NL)]
wq_t = [ops.xattn_tile_weight(w) for w in wq]
wo_t = [ops.xattn_tile_weight(w) for w in wo]
bq, bo = rnd(D, sc=0.02), rnd(D, sc=0.02)
ln_g, ln_b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
h32 = rnd(F * S, D)
h16 = h32.to(dt)
vt = torch.empty(NL * D, ops.pad64(F * N), device=dev, dtype=dt)
qmap = (K, S, 0, 1)
ctxq = torch.zeros(F * K, D, device=dev, dtype=dt)
t32 = torch.empty(F * S, D, device=dev, dtype=torch.float32)


def fused():
    k = ops.gemm(enc, wk, bk)
    ops.gemm(wv, enc, out=vt, c_pad8=True)
    for j in range(NL):
        ops.qformer_xattn(h16, h32, F, K, S, wq_t[j], bq, wo_t[j], bo, k[:, j * D:(j + 1) * D], vt[j * D:(j + 1) * D],
                          bv[j * D:(j + 1) * D], N, ln_g, ln_b, 1e-12, D, heads, 0.125)


def unfused():
    kv = ops.gemm(enc, wkv, bkv)
    ld = kv.stride(0)
    for j in range(NL):
        cq = ops.gemm(h16, wq[j], bq, M=F * K, a_map=qmap)
        ops.attention(cq, kv[:, j * 2 * D:j * 2 * D + D], kv[:, j * 2 * D + D:(j + 1) * 2 * D], ctxq, F, heads, 64, K, N, 0.125,
                      K * cq.stride(0), N * ld, N * ld, K * ctxq.stride(0))
        ops.gemm(ctxq, wo[j], bo, res=h32, r_map=qmap, out=t32, out_f32=True, M=F * K)
        ops.layernorm(t32, ln_g, ln_b, 1e-12, D, dt, y16=h16, y32=h32, rows=F * K, y_map=qmap)


def out_fused(ev=None):
    """the default form (xattn_mode 1): q GEMM and tdc_attention as launches, output projection + residual + LayerNorm fused"""
    def t(kind, fn):
        if ev is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record()
        ev.append((kind, e0, e1))
        return r
    kv = t("kv gemm", lambda: ops.gemm(enc, wkv, bkv))
    ld = kv.stride(0)
    for j in range(NL):
        cq = t("q gemm", lambda: ops.gemm(h16, wq[j], bq, M=F * K, a_map=qmap))
        t("attention", lambda: ops.attention(cq, kv[:, j * 2 * D:j * 2 * D + D], kv[:, j * 2 * D + D:(j + 1) * 2 * D], ctxq, F, heads,
                                             64, K, N, 0.125, K * cq.stride(0), N * ld, N * ld, K * ctxq.stride(0)))
        t("out kernel", lambda: ops.qformer_xattn_out(h16, None, F, K, S, ctxq, wo_t[j], bo, ln_g, ln_b, 1e-12, D, heads, res16=True))


def parts():
    """per-kernel times of the fused form"""
    ev = []

    def t(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record()
        ev.append((e0, e1))
        return r
    k = t(lambda: ops.gemm(enc, wk, bk))
    t(lambda: ops.gemm(wv, enc, out=vt, c_pad8=True))
    for j in range(NL):
        t(lambda: ops.qformer_xattn(h16, h32, F, K, S, wq_t[j], bq, wo_t[j], bo, k[:, j * D:(j + 1) * D], vt[j * D:(j + 1) * D],
                                    bv[j * D:(j + 1) * D], N, ln_g, ln_b, 1e-12, D, heads, 0.125))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]


flop = F * (2.0 * N * H * 2 * D * NL + NL * (4.0 * K * D * D + 4.0 * K * N * D))
for name, fn in (("fused", fused), ("un-fused", unfused), ("out-fused", out_fused)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 10
    for _ in range(R):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / R * 1e3
    print("%-9s F=%d K=%d N=%d %s: %.3f ms per block pass = %.1f TFLOP/s = %.1f %% of 2.5 PFLOP/s (%.2f GFLOP per frame)"
          % (name, F, K, N, str(dt)[6:], ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0, flop / F / 1e9), flush=True)
p = parts()
print("fused parts: key GEMM %.3f ms, value^T GEMM %.3f ms, xattn kernel %s ms" % (p[0], p[1], " ".join("%.3f" % x for x in p[2:])))
ev = []
out_fused(ev)
torch.cuda.synchronize()
by = {}
for kind, a, b in ev:
    by.setdefault(kind, []).append(a.elapsed_time(b))
print("out-fused parts: " + ", ".join("%s %s ms" % (k, " ".join("%.3f" % x for x in v)) for k, v in by.items()))
