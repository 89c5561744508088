"""CPU: why block-scaled (MX, one E8M0 scale per 32 values) operands do not make the e4m3 tower path "parity grade".

VERDICT round 1 asked for MX operands on the strength of "tower RMS error at least halved vs per-row scales".  The error of an
e4m3 GEMM on LayerNorm-like data is set by the format's 3 mantissa bits (relative step 2^-3 ... 2^-4 per value), not by the
granularity of the scale: a finer scale only helps values that would otherwise fall below the format's normal range, and with
per-row / per-tensor scales a row of LayerNorm output (|x| <~ 5 sigma, max / rms ~ 4) sits ~13 binades above e4m3's smallest
normal.  This test measures it: per-row, per-32-block (power-of-two, as MX prescribes) and per-32-block (exact) scales give the
same relative GEMM error within a few percent - so the remaining fp8 levels stay on row scales, and config 5 is quoted with its
measured error, never as a parity result."""
import math

import torch


def q8(x, scale):
    """e4m3 round trip of x / scale (scale broadcastable), values clamped to the format's +-448"""
    y = (x / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()
    return y * scale


def quantise(x, mode):
    M, K = x.shape
    if mode == "row":
        s = x.abs().amax(1, keepdim=True) / 448.0
        return q8(x, s)
    xb = x.view(M, K // 32, 32)
    s = xb.abs().amax(2, keepdim=True) / 448.0
    if mode == "mx":                       # E8M0: power-of-two scales, rounded up so that nothing overflows
        s = torch.exp2(torch.ceil(torch.log2(s.clamp_min(1e-30))))
    return q8(xb, s).view(M, K)


def test_block_scales_do_not_lower_the_e4m3_error_floor():
    g = torch.Generator().manual_seed(0)
    M, N, K = 256, 384, 1152
    x = torch.randn(M, K, generator=g)
    x = torch.nn.functional.layer_norm(x, (K,)) * (1 + 0.1 * torch.randn(K, generator=g))       # LayerNorm output, gamma ~ 1
    w = torch.randn(N, K, generator=g) * 0.02
    ref = x @ w.t()
    errs = {}
    for mode in ("row", "mx", "block"):
        xq = quantise(x, mode)
        wq = quantise(w, "row" if mode == "row" else mode)
        errs[mode] = ((xq @ wq.t() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print("relative RMS error of an e4m3 x e4m3 GEMM, K = 1152: per-row scales %.4f | MX (E8M0 per 32) %.4f | exact per-32 %.4f"
          % (errs["row"], errs["mx"], errs["block"]))
    # 3 mantissa bits on both operands: ~3.5-4.5 % - whatever the scale granularity
    assert 0.02 < errs["row"] < 0.06
    assert errs["mx"] > 0.8 * errs["row"] and errs["block"] > 0.8 * errs["row"], errs
    # ... whereas one more mantissa bit on one operand (e.g. a bf16 activation against e4m3 weights) is what halves it
    xq = x.bfloat16().float()
    e_mixed = (((xq @ quantise(w, "row").t()) - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    assert e_mixed < 0.8 * errs["row"]
    assert math.isfinite(e_mixed)
