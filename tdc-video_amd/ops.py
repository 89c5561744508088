"""torch-tensor wrappers over the C ABI (include/tdc_hip.h).  torch is used for device memory and streams only; every
computation below runs in libtdc_hip.so.  Shapes are validated on the host before a launch (a faulting kernel can
take the whole GPU node down)."""
import ctypes as C
import math

import torch

from . import lib as L


# ---- launch profiler (tdc_profile_* of the C ABI): events are recorded INSIDE the library, around every tdc_gemm /
# tdc_attention / tdc_layernorm / tdc_qformer_xattn launch, whether it comes from a wrapper below or from inside a composite
# (tdc_vit_fwd, tdc_connector_fwd, tdc_qformer_fwd): bench.py's profiled step runs the same host path as its timed steps.
REAL_NK = {}      # weight data_ptr -> un-padded (n, k) of a prepared weight (weights.py), for algorithmic FLOP counts


def register_real_nk(w, n, k):
    """remember the un-padded dims of a prepared weight for the profiler's FLOP counts; the entry dies with the tensor (a freed
    weight's address is reused by later allocations: a stale entry would mis-count whatever GEMM reads that address as its W)"""
    import weakref
    w._real_nk = (n, k)
    ptr = w.data_ptr()
    REAL_NK[ptr] = (n, k)
    weakref.finalize(w, REAL_NK.pop, ptr, None)


def profile_start(max_records=1 << 15):
    L.check(L.load().tdc_profile_start(int(max_records)), "tdc_profile_start")


def profile_tag(tag):
    return L.load().tdc_profile_tag(int(tag))


def profile_stop(max_records=1 << 15):
    """-> list of dicts, one per launch in launch order: kind ('gemm' | 'attn' | 'ln' | 'xattn'), tag, ms, M, N, K, act, res,
    out_f32, flops (GEMMs: from the un-padded weight dims when the weight is registered)."""
    recs = (L.ProfRec * max_records)()
    n = L.load().tdc_profile_stop(recs, max_records)
    if n < 0:
        raise L.TdcHipError("tdc_profile_stop failed with code %d" % n)
    assert n <= max_records, "profile table too small: %d launches" % n
    kinds = {L.PROF_GEMM: "gemm", L.PROF_ATTN: "attn", L.PROF_LN: "ln", L.PROF_XATTN: "xattn"}
    out = []
    for r in recs[:n]:
        fl = r.flops
        if r.kind == L.PROF_GEMM and r.W in REAL_NK:
            rn, rk = REAL_NK[r.W]
            if rn <= r.N and rk <= r.K:         # (a stale pointer whose address an activation buffer reuses does not qualify)
                fl = 2.0 * r.M * rn * rk
        out.append(dict(kind=kinds.get(r.kind, "?"), tag=r.tag, ms=r.ms, M=r.M, N=r.N, K=r.K, act=r.act, res=r.res,
                        out_f32=r.out_f32, flops=fl))
    return out


def _dt(t):
    if t.dtype == torch.float16:
        return L.F16
    if t.dtype == torch.bfloat16:
        return L.BF16
    raise TypeError("16-bit tensor expected, got %s" % t.dtype)


def _dtcode(dtype):
    return L.F16 if dtype == torch.float16 else L.BF16


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk2d(t, name):
    assert t.is_cuda and t.dim() == 2 and t.stride(1) == 1, "%s must be a 2-D row-major CUDA tensor" % name


def pad64(n):
    return (n + 63) // 64 * 64


def _map(m):
    if m is None:
        return L.RowMap(0, 0, 0, 0)
    return L.RowMap(*m)


def _map_max(m, M):
    """largest row index a row map produces for m in [0, M)."""
    if m is None:
        return M - 1
    seg, stride, off, inner = m

    def f(i):
        return (i // seg) * stride + off + (i % seg) * inner
    cand = [M - 1, min(seg - 1, M - 1)]
    if (M // seg) * seg - 1 >= 0:
        cand.append((M // seg) * seg - 1)
    return max(f(i) for i in cand)


FP8_DTYPES = (torch.uint8, torch.float8_e4m3fn)


def gemm(a, w, bias=None, act=L.ACT_NONE, res=None, out=None, out_f32=False, M=None, a_map=None, c_map=None,
         r_map=None, out_rows=None, x16=None, ln_part=None, ln_stats=None, ln_c1=None, out_dtype=None, out_stats=None,
         out_w2max=0.0, out_bmax=0.0, out_wscale=1.0, c_pad8=False):
    """out[c_map(m)] = act(a[a_map(m)] @ w.T + bias) + res[r_map(m)].  a [Ra, lda], w [N, K] (both 16-bit).
    LayerNorm fusion (include/tdc_hip.h): x16 / ln_part = producer outputs (16-bit copy of the fp32 result, per-slot
    (mean, M2) partials [N/64, M, 2]); ln_stats [M, 2] / ln_c1 [N] = consumer inputs (a = raw rows, w = folded weight).
    fp8: a and w hold OCP e4m3 bytes (uint8 / float8_e4m3fn tensors); `out_dtype` names the 16-bit output type and the
    dequantisation scales come in through ln_stats[m] = (row norm bound, s_a[m] * s_w), ln_c1 = 0.  out_stats [M, 2]
    (with `out` an e4m3 byte tensor): the output leaves as e4m3 with analytic per-row scales (tdc_gemm_desc.out_fp8)."""
    _chk2d(a, "a"); _chk2d(w, "w")
    N, K = w.shape
    fp8 = a.dtype in FP8_DTYPES
    if fp8:
        assert w.dtype in FP8_DTYPES and K % 128 == 0 and out_dtype in (torch.float16, torch.bfloat16)
        assert ln_stats is not None, "fp8 operands carry their scales in ln_stats"
        # a residual only as the read-modify-write of the residual stream: fp32, or 16-bit of type `out_dtype`
        assert (res is None and not out_f32) or (out_f32 and res is not None and res.dtype == torch.float32) or \
            (not out_f32 and res is not None and res.dtype == out_dtype and out is not None and out.dtype == out_dtype)
    else:
        out_dtype = a.dtype
    assert (fp8 or a.dtype == w.dtype) and K % 64 == 0 and a.shape[1] >= K, (a.shape, w.shape)
    if M is None:
        M = a.shape[0]
    assert _map_max(a_map, M) < a.shape[0], "a_map out of range"
    n_out = N // 2 if act == L.ACT_SWIGLU else N
    if out is None:
        rows = out_rows if out_rows is not None else M
        out = torch.empty(rows, n_out, device=a.device,
                          dtype=torch.uint8 if out_stats is not None else torch.float32 if out_f32 else out_dtype)
    _chk2d(out, "out")
    assert out.shape[1] >= n_out and _map_max(c_map, M) < out.shape[0], "out too small"
    out8 = out_stats is not None
    if out8:
        assert fp8 and out.dtype in FP8_DTYPES and res is None and not out_f32 and c_map is None
        assert out_stats.dtype == torch.float32 and out_stats.is_contiguous() and out_stats.numel() >= 2 * M
    else:
        assert (out.dtype == torch.float32) == bool(out_f32)
    d = L.GemmDesc()
    if not fp8 and not out_f32 and out.dtype != out_dtype:
        # C (and a 16-bit res) of the other 16-bit type: tdc_gemm_desc.c16_dtype_p1 (an fp16 residual stream under bf16 operands)
        assert out.dtype in (torch.float16, torch.bfloat16) and act == L.ACT_NONE
        assert res is not None, "a 16-bit output of the other 16-bit type exists only on the residual paths (tdc_gemm refuses it too)"
        d.c16_dtype_p1 = _dtcode(out.dtype) + 1
    d.A, d.lda = a.data_ptr(), a.stride(0)
    d.W, d.ldw = w.data_ptr(), w.stride(0)
    d.C, d.ldc = out.data_ptr(), out.stride(0)
    d.bias = bias.data_ptr() if bias is not None else None
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= N and bias.is_contiguous()
    if res is not None:
        _chk2d(res, "res")
        assert res.shape[1] >= N and _map_max(r_map, M) < res.shape[0], "res too small"
        d.res, d.ldres = res.data_ptr(), res.stride(0)
        d.res_f32 = 1 if res.dtype == torch.float32 else 0
        if not d.res_f32:
            assert res.dtype == (a.dtype if out_f32 else out.dtype), \
                "a 16-bit residual has the type of the 16-bit output (of the operands when the output is fp32)"
    d.M, d.N, d.K = M, N, K
    d.dtype, d.out_f32, d.act = _dtcode(out_dtype), int(out_f32), act
    d.in_fp8 = int(fp8)
    d.a_map, d.c_map, d.r_map = _map(a_map), _map(c_map), _map(r_map)
    if x16 is not None:
        _chk2d(x16, "x16")
        assert ln_part is not None and out_f32 and res is not None and N % 64 == 0 and c_map is None and r_map is None
        assert x16.dtype == a.dtype and x16.shape[0] >= M and x16.shape[1] >= N
        assert ln_part.dtype == torch.float32 and ln_part.is_contiguous() and ln_part.numel() >= M * (N // 64) * 2
        d.x16, d.ldx16, d.ln_part = x16.data_ptr(), x16.stride(0), ln_part.data_ptr()
    elif ln_part is not None:
        # the fold's producer over a 16-bit residual stream: only the per-slot partials leave (the consumer reads the stream)
        assert not out_f32 and res is not None and res.dtype == out.dtype and N % 64 == 0 and c_map is None and r_map is None
        assert act == L.ACT_NONE and ln_stats is None and not fp8
        assert ln_part.dtype == torch.float32 and ln_part.is_contiguous() and ln_part.numel() >= M * (N // 64) * 2
        d.ln_part = ln_part.data_ptr()
    if ln_stats is not None:
        assert ln_c1 is not None and a_map is None and (fp8 or (not out_f32 and res is None))
        assert ln_stats.dtype == torch.float32 and ln_stats.is_contiguous() and ln_stats.numel() >= 2 * M
        assert ln_c1.dtype == torch.float32 and ln_c1.is_contiguous() and ln_c1.numel() >= N
        d.ln_stats, d.ln_c1 = ln_stats.data_ptr(), ln_c1.data_ptr()
    if out8:
        d.out_fp8, d.out_stats = 1, out_stats.data_ptr()
        d.out_w2max, d.out_bmax, d.out_wscale = float(out_w2max), float(out_bmax), float(out_wscale)
    if c_pad8:      # rows of `out` writable up to round_up(N, 8) columns (tdc_gemm_desc.c_pad8)
        assert bias is None and res is None and not out_f32 and act == L.ACT_NONE and out.shape[1] >= (N + 7) // 8 * 8
        d.c_pad8 = 1
    L.check(L.load().tdc_gemm(C.byref(d), _stream()), "tdc_gemm")
    return out


def ln_finalize(ln_part, slots, rows, eps, stats=None):
    """per-slot (mean, M2) partials [slots, rows, 2] of a producer GEMM -> stats [rows, 2] = (mean, rstd)."""
    assert ln_part.dtype == torch.float32 and ln_part.is_contiguous() and ln_part.numel() >= rows * slots * 2
    if stats is None:
        stats = torch.empty(rows, 2, device=ln_part.device, dtype=torch.float32)
    assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.numel() >= 2 * rows
    L.check(L.load().tdc_ln_finalize(_ptr(ln_part), slots, rows, eps, _ptr(stats), _stream()), "tdc_ln_finalize")
    return stats


def quantize_rows_fp8(x, cols, wscale, y8=None, stats=None):
    """x [rows, ld] 16-bit -> (y8 [rows, round_up(cols, 128)] e4m3 bytes with per-row scales, stats [rows, 2] = (0, s_a *
    wscale)): the A operand + ln_stats of an fp8-operand gemm whose input is not produced by a LayerNorm."""
    _chk2d(x, "x")
    rows = x.shape[0]
    assert x.shape[1] >= cols and cols % 8 == 0
    if y8 is None:
        y8 = torch.empty(rows, (cols + 127) // 128 * 128, device=x.device, dtype=torch.uint8)
    if stats is None:
        stats = torch.empty(rows, 2, device=x.device, dtype=torch.float32)
    _chk2d(y8, "y8")
    assert y8.dtype in FP8_DTYPES and y8.shape[0] >= rows and y8.shape[1] >= cols
    assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.numel() >= 2 * rows
    L.check(L.load().tdc_quantize_rows_fp8(_ptr(x), x.stride(0), rows, cols, _dt(x), _ptr(y8), y8.stride(0),
                                           _ptr(stats), float(wscale), _stream()), "tdc_quantize_rows_fp8")
    return y8, stats


def layernorm(x, gamma, beta, eps, cols, dtype, y16=None, y32=None, add=None, add_period=0, add_mode=0,
              want16=True, want32=False, rows=None, x_map=None, y_map=None, y8=None, y8_stats=None, y8_wscale=1.0,
              x16_kernel=False):
    """LayerNorm over the first `cols` columns of x [rows, ld] (fp32 or 16-bit) -> (y16, y32).  y8 [rows, ld] uint8 +
    y8_stats [rows, 2]: e4m3 output with per-row scales for an fp8-operand GEMM (include/tdc_hip.h)."""
    if y8 is not None:
        want16 = want16 and y16 is not None
    _chk2d(x, "x")
    rows = x.shape[0] if rows is None else rows
    ld = pad64(cols)
    if want16 and y16 is None:
        y16 = torch.empty(rows, ld, device=x.device, dtype=dtype)
    if want32 and y32 is None:
        y32 = torch.empty(rows, ld, device=x.device, dtype=torch.float32)
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() >= cols
    assert x.shape[1] >= cols and _map_max(x_map, rows) < x.shape[0]
    ymax = _map_max(y_map, rows)
    d = L.LnDesc()
    d.x_map, d.y_map = _map(x_map), _map(y_map)
    d.x, d.ldx, d.x_f32 = x.data_ptr(), x.stride(0), int(x.dtype == torch.float32)
    if x.dtype != torch.float32 and y8 is not None and y16 is None and y32 is None:
        dtype = x.dtype        # e4m3 rows out of a 16-bit stream: `dtype` only names the type of x (no 16-bit output is written)
    elif x.dtype != torch.float32 and (x.dtype != dtype or x16_kernel):
        # 16-bit input of its own type through the 16-bit-to-16-bit kernel (tdc_ln_desc.x_dtype_p1)
        assert y32 is None and y8 is None and add is None and cols % 8 == 0
        d.x_dtype_p1 = _dtcode(x.dtype) + 1
    if y16 is not None:
        assert y16.shape[0] > ymax and y16.shape[1] >= cols and y16.dtype == dtype
        d.y16, d.ldy16 = y16.data_ptr(), y16.stride(0)
    if y32 is not None:
        assert y32.shape[0] > ymax and y32.shape[1] >= cols and y32.dtype == torch.float32
        d.y32, d.ldy32 = y32.data_ptr(), y32.stride(0)
    d.gamma, d.beta, d.eps = gamma.data_ptr(), beta.data_ptr(), eps
    if add is not None:
        assert add.dtype == torch.float32 and add.dim() == 2 and add.shape[1] >= cols
        d.add, d.ldadd, d.add_period, d.add_mode = add.data_ptr(), add.stride(0), add_period, add_mode
        assert add.shape[0] >= (4 if add_mode == 1 else add_period)
    d.rows, d.cols, d.dtype = rows, cols, _dtcode(dtype)
    if y8 is not None:
        _chk2d(y8, "y8")
        assert y8.dtype in FP8_DTYPES and y8.shape[0] > ymax and y8.shape[1] >= cols
        assert y8_stats is not None and y8_stats.dtype == torch.float32 and y8_stats.is_contiguous()
        assert y8_stats.numel() >= 2 * (ymax + 1)
        d.y8, d.ldy8, d.y8_stats, d.y8_wscale = y8.data_ptr(), y8.stride(0), y8_stats.data_ptr(), float(y8_wscale)
    L.check(L.load().tdc_layernorm(C.byref(d), _stream()), "tdc_layernorm")
    return y16, y32


def attention(q, k, v, out, batch, heads, head_dim, sq, sk, scale, q_bs, k_bs, v_bs, o_bs, bias=None, gate=None,
              key_mask=None, form=0):
    """q/k/v/out: 2-D views [tokens, ld] whose element (b, s, h, c) sits at base + b*bs + s*stride(0) + h*head_dim + c.
    Tensors may be column-offset views of a fused QKV buffer.  bias fp32 [heads, sq, sk] + gate fp32 [batch*sq, >=heads]
    add gate[b*sq+q, h] * bias[h, q, k] to the scaled scores (BEATs gated relative position bias); key_mask uint8 / bool
    [batch, sk] (biased form only): non-zero = key excluded from the softmax (key padding mask)."""
    for t in (q, k, v, out):
        assert t.is_cuda and t.dim() == 2 and t.stride(1) == 1
    assert q.dtype == k.dtype == v.dtype == out.dtype
    assert heads * head_dim <= q.shape[1] and heads * head_dim <= out.shape[1]

    def last(t, bs, s):
        return (batch - 1) * bs + (s - 1) * t.stride(0) + heads * head_dim
    for t, bs, s in ((q, q_bs, sq), (k, k_bs, sk), (v, v_bs, sk), (out, o_bs, sq)):
        span = (t.shape[0] - 1) * t.stride(0) + t.shape[1]
        assert last(t, bs, s) <= span, "attention view out of range"
    d = L.AttnDesc()
    d.q, d.k, d.v, d.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    d.q_bs, d.k_bs, d.v_bs, d.o_bs = q_bs, k_bs, v_bs, o_bs
    d.q_rs, d.k_rs, d.v_rs, d.o_rs = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    d.batch, d.heads, d.head_dim, d.sq, d.sk = batch, heads, head_dim, sq, sk
    d.scale, d.dtype, d.form = scale, _dt(q), form     # form 1: the 16x16x32 kernels for every shape (tdc_attn_desc.form)
    if bias is not None:
        assert gate is not None and bias.is_cuda and gate.is_cuda and bias.dtype == gate.dtype == torch.float32
        assert bias.dim() == 3 and bias.is_contiguous() and tuple(bias.shape) == (heads, sq, sk) and sk % 4 == 0
        assert gate.dim() == 2 and gate.stride(1) == 1 and gate.shape[0] >= batch * sq and gate.shape[1] >= heads
        assert head_dim <= 64 and head_dim % 8 == 0
        d.bias, d.bias_hs, d.bias_rs = bias.data_ptr(), sq * sk, sk
        d.gate, d.gate_rs = gate.data_ptr(), gate.stride(0)
    if key_mask is not None:
        assert bias is not None, "key_mask: biased form only"
        assert key_mask.is_cuda and key_mask.dtype in (torch.uint8, torch.bool) and key_mask.is_contiguous()
        assert tuple(key_mask.shape) == (batch, sk)
        d.key_mask, d.key_mask_bs = key_mask.data_ptr(), sk
    L.check(L.load().tdc_attention(C.byref(d), _stream()), "tdc_attention")
    return out


def px_kind(px, dtype):
    """the px_f32 argument of tdc_im2col / tdc_vit_fwd: 0 = pixels of the towers' 16-bit type, 1 = fp32, 2 = the other one"""
    if px.dtype == torch.float32:
        return 1
    assert px.dtype in (torch.float16, torch.bfloat16) and dtype in (torch.float16, torch.bfloat16)
    return 0 if px.dtype == dtype else 2


def im2col(px, patch, dtype):
    B, Cc, H, W = px.shape
    assert Cc == 3 and px.is_contiguous() and px.is_cuda
    gh, gw = H // patch, W // patch
    ldp = pad64(3 * patch * patch)
    out = torch.empty(B * gh * gw, ldp, device=px.device, dtype=dtype)
    f32 = px_kind(px, dtype)
    L.check(L.load().tdc_im2col(_ptr(px), f32, _ptr(out), ldp, B, H, W, patch, _dtcode(dtype), _stream()),
            "tdc_im2col")
    return out, gh, gw


def set_rows(x32, B, S, row, vec):
    assert x32.dtype == torch.float32 and x32.dim() == 2 and x32.shape[0] >= B * S and vec.numel() >= x32.shape[1]
    L.check(L.load().tdc_set_rows(_ptr(x32), x32.stride(0), B, S, row, _ptr(vec), _stream()), "tdc_set_rows")


def set_rows16(x16, B, S, row, vec):
    assert x16.dtype in (torch.float16, torch.bfloat16) and x16.dim() == 2 and x16.shape[0] >= B * S
    assert vec.dtype == torch.float32 and vec.numel() >= x16.shape[1]
    L.check(L.load().tdc_set_rows16(_ptr(x16), x16.stride(0), B, S, row, _ptr(vec), _dtcode(x16.dtype), _stream()),
            "tdc_set_rows16")


def bilinear_tables(n_in, n_out, device):
    """index / weight tables of F.interpolate(bilinear, align_corners=False) along one axis."""
    i0, i1, fr = [], [], []
    scale = n_in / n_out
    for o in range(n_out):
        src = max((o + 0.5) * scale - 0.5, 0.0)
        a = int(math.floor(src))
        i0.append(a)
        i1.append(min(a + 1, n_in - 1))
        fr.append(src - a)
    return (torch.tensor(i0, dtype=torch.int32, device=device), torch.tensor(i1, dtype=torch.int32, device=device),
            torch.tensor(fr, dtype=torch.float32, device=device))


def resample_tokens(x, B, tok_off, n_in, n_out, cols, dtype, tables, out_dtype=None):
    """dtype: 16-bit type of x (when x is not fp32); out_dtype: 16-bit type of the result (default: dtype)."""
    _chk2d(x, "x")
    assert x.shape[0] >= B * (tok_off + n_in * n_in) and x.shape[1] >= cols
    ldy = pad64(cols)
    out_dtype = dtype if out_dtype is None else out_dtype
    y = torch.empty(B * n_out * n_out, ldy, device=x.device, dtype=out_dtype)
    i0, i1, fr = tables
    assert i0.numel() == n_out and int(i1.max()) < n_in
    L.check(L.load().tdc_resample_tokens(_ptr(x), int(x.dtype == torch.float32), x.stride(0), tok_off, n_in, _ptr(y),
                                         ldy, n_out, _ptr(i0), _ptr(i1), _ptr(fr), B, cols, _dtcode(dtype),
                                         _dtcode(out_dtype), _stream()),
            "tdc_resample_tokens")
    return y


def frame_cossim(f, T, n):
    """f: 16-bit [T*tokens, ld] viewed as [T, n] -> sims fp32 [T-1]."""
    assert f.is_cuda and f.is_contiguous() and f.numel() >= T * n and n % 8 == 0 and T >= 2
    lib = L.load()
    scratch = torch.empty(lib.tdc_frame_cossim_scratch_floats(T), device=f.device, dtype=torch.float32)
    sims = torch.empty(T - 1, device=f.device, dtype=torch.float32)
    L.check(lib.tdc_frame_cossim(_ptr(f), n, T, _ptr(sims), _ptr(scratch), _dt(f), _stream()), "tdc_frame_cossim")
    return sims


def token_mean(x, B, P):
    _chk2d(x, "x")
    assert x.shape[0] >= B * P and x.is_contiguous()
    y = torch.empty(B, x.shape[1], device=x.device, dtype=x.dtype)
    L.check(L.load().tdc_token_mean(_ptr(x), P, x.stride(0), _ptr(y), B, _dt(x), _stream()), "tdc_token_mean")
    return y


def adaptive_pool_tokens(x, N, K, B, src_row=None, frame_rows=None):
    _chk2d(x, "x")
    assert x.is_contiguous()
    frame_rows = N if frame_rows is None else frame_rows
    assert frame_rows >= N
    if src_row is not None:
        assert src_row.dtype == torch.int32 and src_row.numel() >= B
        assert int(src_row.max()) * frame_rows + N <= x.shape[0] and int(src_row.min()) >= 0
    else:
        assert (B - 1) * frame_rows + N <= x.shape[0]
    y = torch.empty(B * K, x.shape[1], device=x.device, dtype=x.dtype)
    L.check(L.load().tdc_adaptive_pool_tokens(_ptr(x), N, frame_rows, x.stride(0), _ptr(y), K, B, _ptr(src_row),
                                              _dt(x), _stream()), "tdc_adaptive_pool_tokens")
    return y


def check_pairs_host(pairs, table_rows):
    """host-side range check of a (table, row) gather map (numpy int array [n, 2]) against the tables' row counts - the
    kernel trusts its indices, and a faulting kernel can take the whole GPU node down."""
    import numpy as np
    p = np.asarray(pairs).reshape(-1, 2)
    if p.shape[0] == 0:
        return
    assert p[:, 0].min() >= 0 and p[:, 0].max() < len(table_rows), "gather table out of range"
    assert p[:, 1].min() >= 0, "negative gather row"
    lim = np.asarray(table_rows, dtype=np.int64)[p[:, 0]]
    assert (p[:, 1] < lim).all(), "gather row out of range"


def gather_rows(tables, src, n, cols, out=None, validated=False):
    """tables: list of <= 4 2-D 16-bit tensors; src int32 [n, 2] (table, row) on the device.  validated=True: the caller
    has range-checked the map on the host (check_pairs_host) - skips the device-side check and its synchronisations."""
    gt = L.GatherTables()
    dtype = tables[0].dtype
    rows_ok = []
    for i, t in enumerate(tables):
        if t.dim() == 1:
            t = t.view(1, -1)
        assert t.is_cuda and t.stride(1) == 1 and t.dtype == dtype and t.shape[1] >= cols
        gt.base[i] = t.data_ptr()
        gt.ld[i] = t.stride(0)
        rows_ok.append(t.shape[0])
    assert src.dtype == torch.int32 and src.is_contiguous() and src.numel() >= 2 * n
    if not validated:
        s = src.view(-1, 2)[:n]
        assert int(s[:, 0].max()) < len(tables) and int(s[:, 0].min()) >= 0
        for i, r in enumerate(rows_ok):
            sel = s[:, 1][s[:, 0] == i]
            if sel.numel():
                assert int(sel.max()) < r and int(sel.min()) >= 0, "gather row out of range"
    if out is None:
        out = torch.empty(n, cols, device=src.device, dtype=dtype)
    assert out.shape[0] >= n and out.shape[1] >= cols and out.stride(1) == 1
    L.check(L.load().tdc_gather_rows(C.byref(gt), _ptr(src), _ptr(out), out.stride(0), n, cols, _dtcode(dtype),
                                     _stream()), "tdc_gather_rows")
    return out


def l2_normalize(x, rows, cols):
    _chk2d(x, "x")
    assert x.shape[0] >= rows and x.shape[1] >= cols
    L.check(L.load().tdc_l2_normalize(_ptr(x), x.stride(0), rows, cols, _dt(x), _stream()), "tdc_l2_normalize")
    return x


def sva_attention(q, kv_list, mask, T, side, r, dim, heads, out=None):
    _chk2d(q, "q")
    nq = T * side * side
    n = side * r
    assert q.shape[0] >= nq and q.shape[1] >= dim
    d = L.SvaAttnDesc()
    d.q, d.ldq = q.data_ptr(), q.stride(0)
    for i, kv in enumerate(kv_list):
        _chk2d(kv, "kv")
        assert kv.shape[0] >= T * n * n and kv.shape[1] >= 2 * dim and kv.dtype == q.dtype
        assert kv.stride(0) == kv_list[0].stride(0)
        d.kv[i] = kv.data_ptr()
    d.ldkv = kv_list[0].stride(0)
    nkv = len(kv_list) * r * r
    assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.numel() >= nq * nkv
    d.mask = mask.data_ptr()
    if out is None:
        out = torch.empty(nq, pad64(dim), device=q.device, dtype=q.dtype)
        if out.shape[1] > dim:
            out[:, dim:].zero_()
    d.out, d.ldo = out.data_ptr(), out.stride(0)
    d.T, d.side, d.r, d.n_towers, d.dim, d.heads, d.dtype = T, side, r, len(kv_list), dim, heads, _dt(q)
    L.check(L.load().tdc_sva_attention(C.byref(d), _stream()), "tdc_sva_attention")
    return out


def qformer_embed(query, qsrc, word, pos, ids, gamma, beta, eps, F, K, cols, dtype):
    _chk2d(query, "query")
    Lt = 0 if ids is None else ids.numel()
    ld = pad64(cols)
    rows = F * (K + Lt)
    h32 = torch.empty(rows, ld, device=query.device, dtype=torch.float32)
    h16 = torch.empty(rows, ld, device=query.device, dtype=dtype)
    assert qsrc.dtype == torch.int32 and qsrc.numel() >= F and (int(qsrc.max()) + 1) * K <= query.shape[0]
    d = L.QEmbedDesc()
    d.query, d.ldq, d.qsrc = query.data_ptr(), query.stride(0), qsrc.data_ptr()
    if Lt:
        assert ids.dtype == torch.int32 and word.dtype == torch.float32 and pos.dtype == torch.float32
        assert int(ids.max()) < word.shape[0] and Lt <= pos.shape[0] and word.stride(0) == pos.stride(0)
        d.word, d.pos, d.ldw, d.ids = word.data_ptr(), pos.data_ptr(), word.stride(0), ids.data_ptr()
    d.Lt = Lt
    d.gamma, d.beta, d.eps = gamma.data_ptr(), beta.data_ptr(), eps
    d.h32, d.h16, d.ld = h32.data_ptr(), h16.data_ptr(), ld
    d.F, d.K, d.cols, d.dtype = F, K, cols, _dtcode(dtype)
    L.check(L.load().tdc_qformer_embed(C.byref(d), _stream()), "tdc_qformer_embed")
    return h32, h16


def qformer_xattn_supported(dim, heads, K, Nenc):
    return bool(L.load().tdc_qformer_xattn_supported(dim, heads, K, Nenc))


def xattn_tile_weight(w):
    """[>= 768, ld] 16-bit nn.Linear weight -> its fragment-major copy for tdc_qformer_xattn (768 * 768 values, flat)."""
    _chk2d(w, "w")
    assert w.shape[0] >= 768 and w.shape[1] >= 768
    out = torch.empty(768 * 768, device=w.device, dtype=w.dtype)
    L.check(L.load().tdc_qformer_xattn_tile_weight(_ptr(w), w.stride(0), _ptr(out), _dt(w), _stream()),
            "tdc_qformer_xattn_tile_weight")
    return out


def qformer_xattn_out(h16, h32, F, K, S, ctx, wo_t, bo, ln_g, ln_b, eps, dim, heads, res16=False):
    """The last third of the block alone (tdc_qformer_xattn with ctx): ctx [F*K, ld] = the attention output of the flat query
    rows -> h16 / h32 query rows = LayerNorm(ctx Wo^T + bo + h).  res16: the residual is h16 and only h16 is written (h32 may
    be None)."""
    for t in (h16, ctx) + (() if (res16 and h32 is None) else (h32,)):
        _chk2d(t, "xattn operand")
    assert h16.dtype == wo_t.dtype == ctx.dtype and h16.shape[0] >= F * S and h16.shape[1] >= dim
    if h32 is not None:
        assert h32.dtype == torch.float32 and h32.shape[0] >= F * S and h16.stride(0) == h32.stride(0)
    else:
        assert res16
    assert ctx.shape[0] >= F * K and ctx.shape[1] >= dim and wo_t.is_cuda and wo_t.is_contiguous() and wo_t.numel() == dim * dim
    for v in (bo, ln_g, ln_b):
        assert v.dtype == torch.float32 and v.is_contiguous() and v.numel() >= dim
    d = L.XattnDesc()
    d.h16, d.h32, d.ldh = h16.data_ptr(), (h32.data_ptr() if h32 is not None else None), h16.stride(0)
    d.F, d.K, d.S = F, K, S
    d.res16 = 1 if res16 else 0
    d.wo, d.bo = wo_t.data_ptr(), bo.data_ptr()
    d.ln_g, d.ln_b, d.eps = ln_g.data_ptr(), ln_b.data_ptr(), eps
    d.dim, d.heads, d.dtype, d.Nenc = dim, heads, _dt(h16), 8
    d.ctx, d.ldctx = ctx.data_ptr(), ctx.stride(0)
    L.check(L.load().tdc_qformer_xattn(C.byref(d), _stream()), "tdc_qformer_xattn")


def qformer_xattn(h16, h32, F, K, S, wq_t, bq, wo_t, bo, k, vt, bv, Nenc, ln_g, ln_b, eps, dim, heads, scale):
    """The Q-Former cross-attention block of one layer in one launch (tdc_qformer_xattn; tdc/Qformer.py:128-130,185-188,
    205-264,285-289): the K query rows of each of the F frames in h16 / h32 [F*S, ld] are replaced by
    LayerNorm(softmax(q k^T * scale) v Wo^T + bo + h), q = h Wq^T + bq.  wq_t / wo_t: xattn_tile_weight() copies of the two
    weights; k [F*Nenc, >= dim] (a column view of the stacked key rows), vt [dim, ldvt] (a row view of the stacked TRANSPOSED
    values, no bias), bv [dim] fp32."""
    for t in (h16, h32, k, vt):
        _chk2d(t, "xattn operand")
    assert h32.dtype == torch.float32 and h16.dtype == wq_t.dtype == wo_t.dtype == k.dtype == vt.dtype
    assert h16.shape[0] >= F * S and h32.shape[0] >= F * S and h16.stride(0) == h32.stride(0) and h16.shape[1] >= dim
    assert wq_t.is_cuda and wo_t.is_cuda and wq_t.is_contiguous() and wo_t.is_contiguous()
    assert wq_t.numel() == dim * dim and wo_t.numel() == dim * dim
    assert k.shape[0] >= F * Nenc and k.shape[1] >= dim and vt.shape[0] >= dim and vt.shape[1] >= F * Nenc
    for v in (bq, bo, ln_g, ln_b) + ((bv,) if bv is not None else ()):
        assert v.dtype == torch.float32 and v.is_contiguous() and v.numel() >= dim
    d = L.XattnDesc()
    d.h16, d.h32, d.ldh = h16.data_ptr(), h32.data_ptr(), h16.stride(0)
    d.F, d.K, d.S = F, K, S
    d.wq, d.bq, d.wo, d.bo = wq_t.data_ptr(), bq.data_ptr(), wo_t.data_ptr(), bo.data_ptr()
    d.k, d.ldk = k.data_ptr(), k.stride(0)
    d.vt, d.ldvt, d.bv = vt.data_ptr(), vt.stride(0), bv.data_ptr() if bv is not None else None
    d.Nenc, d.ln_g, d.ln_b, d.eps = Nenc, ln_g.data_ptr(), ln_b.data_ptr(), eps
    d.dim, d.heads, d.scale, d.dtype = dim, heads, scale, _dt(h16)
    L.check(L.load().tdc_qformer_xattn(C.byref(d), _stream()), "tdc_qformer_xattn")


def fbank(wav, tables, dtype, want_plain=False, mean=15.41663, std=6.55582):
    """wav [B, n] fp16/fp32 on the GPU -> (patches [B*(m//16)*8, 256] `dtype`, plain fp32 [B, m, 128] or None, m).
    tables = (window, twiddle, banks, range) device tensors (beats.fbank_tables)."""
    assert wav.is_cuda and wav.dim() == 2 and wav.stride(1) == 1 and wav.dtype in (torch.float16, torch.float32)
    B, n = wav.shape
    m = L.load().tdc_fbank_frames(n)
    assert m >= 16, "audio window shorter than one 16-frame patch row"
    window, tw, banks, rng = tables
    assert window.numel() == 400 and tw.numel() == 512 and tuple(banks.shape) == (128, 257) and tuple(rng.shape) == (128, 2)
    for t in (window, tw, banks):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    assert rng.is_cuda and rng.dtype == torch.int32 and rng.is_contiguous()
    patches = torch.empty(B * (m // 16) * 8, 256, device=wav.device, dtype=dtype)
    plain = torch.empty(B, m, 128, device=wav.device, dtype=torch.float32) if want_plain else None
    L.check(L.load().tdc_fbank(_ptr(wav), 1 if wav.dtype == torch.float32 else 0, n, wav.stride(0), B, _ptr(window),
                               _ptr(tw), _ptr(banks), _ptr(rng), _ptr(plain) if plain is not None else None,
                               _ptr(patches), 256, _dtcode(dtype), mean, 1.0 / (2.0 * std), _stream()), "tdc_fbank")
    return patches, plain, m


def relpos_gate(q, rows, heads, head_dim, w2, b2, grep_a, out=None):
    """q: [rows, ld] 16-bit view (un-scaled q_proj output) -> gate fp32 [rows, heads]."""
    assert q.is_cuda and q.dim() == 2 and q.stride(1) == 1 and q.shape[0] >= rows and q.shape[1] >= heads * head_dim
    for t, n in ((w2, 2 * head_dim), (b2, 2), (grep_a, heads)):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n
    if out is None:
        out = torch.empty(rows, heads, device=q.device, dtype=torch.float32)
    L.check(L.load().tdc_relpos_gate(_ptr(q), q.stride(0), rows, heads, head_dim, _ptr(w2), _ptr(b2), _ptr(grep_a),
                                     _ptr(out), out.stride(0), _dt(q), _stream()), "tdc_relpos_gate")
    return out
