"""The one-wave-per-SIMD attention prototype (tools/attention_pw/attention_pw.hip, its own libtdc_attn_pw.so - build it with
tools/attention_pw/build.sh, which also audits the generated code) against the library's tower kernel: correctness on the
cases round 4's test suite held (layout by exact selections, fp32 SDPA, a late running-maximum jump, ragged tile / block
edges, both 16-bit types), then time at the tower shape and at a long sequence (where the per-item prologue is amortised).
python tools/debug_attn_pw.py"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import ops, lib as L  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

PW = C.CDLL(os.path.join(HERE, "attention_pw", "libtdc_attn_pw.so"))
PW.tdc_attn_pw_run.restype = C.c_int
PW.tdc_attn_pw_run.argtypes = [C.POINTER(L.AttnDesc), C.c_void_p]


def attention_pw(q, k, v, out, batch, heads, sq, sk, scale, q_bs, k_bs, v_bs, o_bs):
    d = L.AttnDesc()
    d.q, d.k, d.v, d.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    d.q_bs, d.k_bs, d.v_bs, d.o_bs = q_bs, k_bs, v_bs, o_bs
    d.q_rs, d.k_rs, d.v_rs, d.o_rs = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    d.batch, d.heads, d.head_dim, d.sq, d.sk = batch, heads, 64, sq, sk
    d.scale, d.dtype = scale, ops._dt(q)
    L.check(PW.tdc_attn_pw_run(C.byref(d), ops._stream()), "tdc_attn_pw_run")
    return out


def check(S):
    B, H, d = 2, 3, 64
    D = H * d
    ld = ops.pad64(D)

    def run(q, k, v, pw=True):
        out = torch.zeros(B * S, ld, device="cuda", dtype=q.dtype)
        if pw:
            attention_pw(q[:, :D], k[:, :D], v[:, :D], out, B, H, S, S, 0.125, S * ld, S * ld, S * ld, S * ld)
        else:
            ops.attention(q[:, :D], k[:, :D], v[:, :D], out, B, H, d, S, S, 0.125, S * ld, S * ld, S * ld, S * ld)
        return out[:, :D].float()
    z = torch.zeros(B * S, ld, device="cuda", dtype=torch.float16)
    rows = torch.arange(B * S, device="cuda").view(-1, 1)
    cols = torch.arange(D, device="cuda").view(1, -1)
    v = torch.zeros_like(z)
    v[:, :D] = ((rows * 7 + cols * 3) % 13).half()
    got = run(z, z, v)                                                   # uniform attention = the mean of V
    ref = v[:, :D].float().view(B, S, D).mean(1, keepdim=True).expand(B, S, D).reshape(B * S, D)
    assert (got - ref).abs().max().item() < 2e-2
    q, k = torch.zeros_like(z), torch.zeros_like(z)
    sel = (torch.arange(S, device="cuda") * 37 + 11) % S
    code = torch.zeros(S, d, device="cuda")
    for bit in range(max(1, (S - 1).bit_length())):
        code[:, bit] = ((torch.arange(S, device="cuda") >> bit) & 1).float() * 2 - 1
    for bi in range(B):
        for hi in range(H):
            k[bi * S:(bi + 1) * S, hi * d:(hi + 1) * d] = code.half()
            q[bi * S:(bi + 1) * S, hi * d:(hi + 1) * d] = (code[sel] * 96.0).half()      # x scale 0.125 = 12
    got = run(q, k, v)                                                   # one-hot attention = a row selection
    ref = v[:, :D].float().view(B, S, D)[:, sel].reshape(B * S, D)
    assert (got - ref).abs().max().item() < 5e-2
    g = torch.Generator(device="cuda").manual_seed(8)
    worst = 0.0
    for dtype in (torch.float16, torch.bfloat16):
        qkv = [torch.randn(B * S, ld, device="cuda", generator=g).to(dtype) for _ in range(3)]
        qkv[0] = (qkv[0].float() * 3).to(dtype)                            # peaky rows: the maximum moves for many tiles
        for hi in range(H):     # a spike late in the sequence for one query of every head: the running maximum jumps in the LAST tiles
            qkv[1][S - 3, hi * d:(hi + 1) * d] = (qkv[0][5, hi * d:(hi + 1) * d].float() * 4).to(dtype)
        a = run(*qkv)
        qf, kf, vf = (t[:, :D].float().view(B, S, H, d).transpose(1, 2) for t in qkv)
        ref = F.scaled_dot_product_attention(qf, kf, vf, scale=0.125).transpose(1, 2).reshape(B * S, D)
        err = (a - ref).abs().max().item()
        assert err < (4e-3 if dtype == torch.float16 else 3e-2), (S, dtype, err)
        b = run(*qkv, pw=False)
        assert (a - b).abs().max().item() < (4e-3 if dtype == torch.float16 else 3e-2)
        worst = max(worst, err)
    print("S=%d ok (max error against fp32 SDPA %.2e)" % (S, worst), flush=True)


for S in (730, 729, 256, 320, 1501):          # (the form needs >= 256 query rows and >= 192 keys)
    check(S)

g = torch.Generator(device="cuda").manual_seed(0)
for (B, H, S) in [(8, 24, 4096), (512, 24, 730)]:
    D = H * 64
    qkv = torch.randn(B * S, ops.pad64(3 * D), device="cuda", generator=g).half()
    ld = qkv.shape[1]
    ref = None
    for pw in (False, True):
        out = torch.zeros(B * S, ops.pad64(D), device="cuda", dtype=qkv.dtype)
        if pw:
            fn = lambda: attention_pw(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], out, B, H, S, S, 0.125, S * ld, S * ld,
                                      S * ld, S * out.stride(0))
        else:
            fn = lambda: ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], out, B, H, 64, S, S, 0.125, S * ld,
                                       S * ld, S * ld, S * out.stride(0))
        fn()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.float().clone()
        ms = timeit(fn)
        print("B=%d H=%d S=%d %s: %8.3f ms %7.1f TFLOP/s  max diff vs the library's kernel %.3e" %
              (B, H, S, "prototype (one wave per SIMD)" if pw else "library (attention32.hip)    ", ms,
               4.0 * B * H * S * S * 64 / ms / 1e9, (out.float() - ref).abs().max().item()), flush=True)
