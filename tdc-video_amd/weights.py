"""Weight preparation: reference-named state dict -> device-resident, padded, fused weights.

Done once per model (like checkpoint loading), never on the per-video path:
  * every nn.Linear weight [out, in] is zero-padded to [pad64(out), pad64(in)] and cast to the 16-bit compute type,
    biases / LayerNorm parameters / position tables stay fp32 (zero-padded);
  * q|k|v projections are stacked into one GEMM; the six Q-Former cross-attention K/V projections (which all read the
    same encoder tokens) are stacked into one [12*768, H] GEMM;
  * DINOv2 LayerScale is folded into the preceding Linear (tdc reference: HF dinov2 Dinov2LayerScale);
  * DINOv2 SwiGLU rows are interleaved (x1_j, x2_j) so the GEMM epilogue can apply silu(x1)*x2 in registers;
  * SVA k/v LayerNorm affines are folded into the k/v Linear (the two LayerNorms of one tower share their statistics);
  * the DINOv2 position table is bicubic-resampled to the patch grid (HF:models/dinov2/modeling_dinov2.py:79-88).
"""
import math

import torch


def pad64(n):
    return (n + 63) // 64 * 64


def _register_real_nk(w, n, k):
    """un-padded dims of a prepared weight: kept on the tensor and, by data pointer, for the library's launch profiler"""
    from . import ops
    ops.register_real_nk(w, n, k)


class Lin:
    """Prepared nn.Linear: w [n_pad, k_pad] 16-bit, b fp32 [n_pad] or None."""
    __slots__ = ("w", "b", "n", "k", "wscale", "zeros", "w2max", "bmax")

    def __init__(self, w, b, n, k, wscale=None, zeros=None, w2max=0.0, bmax=0.0):
        self.w2max, self.bmax = w2max, bmax         # fp8 weights: max row 2-norm of w (e4m3 units), max |b| (out_fp8 bound)
        self.w, self.b, self.n, self.k = w, b, n, k
        self.wscale, self.zeros = wscale, zeros     # fp8 weights (to_fp8): per-tensor scale, zero ln_c1 vector


def make_lin(W, b, dtype, dev, row_scale=None, col_scale=None, col_shift=None, n_pad=None, k_pad=None):
    """W [n,k].  Effective op: y = row_scale * (W @ (col_scale * x + col_shift) + b).  The folding arithmetic runs in
    fp32 (one-time weight preparation) on `dev` when that is a GPU - host weights are uploaded first: padding and converting
    1.5 G parameters is seconds of host time otherwise - else on the device W lives on; the result is cast and moved to `dev`."""
    W = W.detach().to(torch.float32)
    if W.device.type == "cpu" and torch.device(dev).type == "cuda":
        W = W.to(dev)
    wd = W.device
    n, k = W.shape
    bias = b.detach().to(torch.float32).to(wd).clone() if b is not None else None
    if col_shift is not None:
        extra = W @ col_shift.detach().to(torch.float32).to(wd)
        bias = extra if bias is None else bias + extra
    if col_scale is not None:
        W = W * col_scale.detach().to(torch.float32).to(wd)[None, :]
    if row_scale is not None:
        rs = row_scale.detach().to(torch.float32).to(wd)
        W = W * rs[:, None]
        if bias is not None:
            bias = bias * rs
    n_pad = pad64(n) if n_pad is None else n_pad
    k_pad = pad64(k) if k_pad is None else k_pad
    Wp = torch.zeros(n_pad, k_pad, dtype=dtype, device=wd)
    Wp[:n, :k] = W.to(dtype)
    bp = None
    if bias is not None:
        bp = torch.zeros(n_pad, dtype=torch.float32, device=wd)
        bp[:n] = bias
        bp = bp.to(dev)
    w = Wp.to(dev).contiguous()
    _register_real_nk(w, n, k)  # algorithmic dims for FLOP accounting (bench.py)
    return Lin(w, bp, n, k)


def stack_lins(parts, dtype, dev, k_pad=None, n_pad=None):
    """parts: list of (W, b, kwargs) stacked along the output dim WITHOUT per-part padding."""
    Ws, bs = [], []
    any_b = any(p[1] is not None or (p[2] or {}).get("col_shift") is not None for p in parts)
    for W, b, kw in parts:
        kw = kw or {}
        l = make_lin(W, b, torch.float32, dev if torch.device(dev).type == "cuda" else W.device, n_pad=W.shape[0],
                     k_pad=W.shape[1], **kw)
        Ws.append(l.w)
        if any_b:
            bs.append(l.b if l.b is not None else torch.zeros(W.shape[0], device=l.w.device))
    Wc = torch.cat(Ws, 0)
    bc = torch.cat(bs, 0) if any_b else None
    return make_lin(Wc, bc, dtype, dev, k_pad=k_pad, n_pad=n_pad)


def to_fp8(lin, dev):
    """16-bit prepared Lin -> fp8-operand Lin: w = e4m3(W / s_w) bytes [n_pad, round_up(k, 128)] with the per-tensor scale
    s_w = max|W| / 448 (tdc_gemm_desc.in_fp8).  The activation side is quantised per row by the LayerNorm kernel
    (tdc_ln_desc.y8), whose y8_stats carry s_a * s_w into the GEMM epilogue."""
    W = lin.w.float()
    amax = float(W.abs().max())
    sw = amax / 448.0 if amax > 0 else 1.0
    n_pad, k_pad = W.shape[0], (W.shape[1] + 127) // 128 * 128
    W8 = torch.zeros(n_pad, k_pad, dtype=torch.uint8, device=W.device)
    Wq = (W / sw).to(torch.float8_e4m3fn)
    W8[:, : W.shape[1]] = Wq.view(torch.uint8)
    w2max = float(Wq.float().norm(dim=1).max())
    bmax = float(lin.b.abs().max()) if lin.b is not None else 0.0
    W8 = W8.to(dev).contiguous()
    _register_real_nk(W8, *getattr(lin.w, "_real_nk", (lin.n, lin.k)))
    return Lin(W8, lin.b, lin.n, lin.k, wscale=sw, zeros=torch.zeros(n_pad, dtype=torch.float32, device=dev),
               w2max=w2max, bmax=bmax)


def fp8_enabled(dim, requested):
    """fp8 (e4m3) operand level of a tower of width `dim` (whole 128-byte K tiles only): 0 = none, 1 = the GEMMs fed by a
    LayerNorm (qkv, fc1: the LayerNorm kernel quantises), 2 = also out-proj / fc2 (their inputs - attention output, MLP
    hidden - go through tdc_quantize_rows_fp8), 3 = as 2 with the MLP hidden written as e4m3 by fc1 itself
    (tdc_gemm_desc.out_fp8; fp8_level3_ok).  `requested`: False / True (= 1) / 1 / 2 / 3."""
    level = int(requested) if requested else 0
    return level if dim % 128 == 0 else 0


def fp8_level3_ok(t):
    """fc1 can write fc2's e4m3 operand directly when its (padded) output width is fc2's padded K."""
    Lr = t.layers[0]
    n_out = Lr.fc1.w.shape[0] // (2 if t.act == "swiglu" else 1)
    return n_out == Lr.fc2.w.shape[1]


def fold_c1(lin):
    """Row sums of a prepared (16-bit, padded) weight, fp32: the `ln_c1` operand of a LayerNorm-folded GEMM
    (include/tdc_hip.h, tdc_gemm_desc).  Summed from the ROUNDED weight so that mean * c1 cancels exactly what the MFMAs
    accumulate."""
    return lin.w.float().sum(1).contiguous()


def ln_fusion_enabled(dim, requested=False):
    """LayerNorm fusion of the towers (VideoEncoder(ln_fuse=True); no environment variable): the pre-LayerNorms fold into the
    neighbouring GEMMs (widths that are a whole number of 64-column slots).  Off by default: measured neutral on MI355X
     - the fold moves the LayerNorm's bytes (an HBM-bound kernel at 6 TB/s) into the GEMM epilogue, which
    is not overlapped with MFMA work; the LayerNorm kernel is also the path with the larger numerical margin.  Kept as a
    tested alternative (bit-identical statistics whichever GEMM kernel computes a row), not as a tuning knob."""
    return dim % 64 == 0 and bool(requested)


def vec32(v, dev, n_pad=None, fill=0.0):
    v = v.detach().to(torch.float32).flatten()
    n_pad = pad64(v.numel()) if n_pad is None else n_pad
    out = torch.full((n_pad,), fill, dtype=torch.float32, device=v.device)
    out[: v.numel()] = v
    return out.to(dev)


def mat32(m, dev, cols_pad=None):
    m = m.detach().to(torch.float32)
    cols_pad = pad64(m.shape[1]) if cols_pad is None else cols_pad
    out = torch.zeros(m.shape[0], cols_pad, dtype=torch.float32, device=m.device)
    out[:, : m.shape[1]] = m
    return out.to(dev).contiguous()


def bicubic_matrix(n_in, n_out, A=-0.75):
    """[n_out, n_in] matrix of F.interpolate(mode='bicubic', align_corners=False) along one axis."""
    def c1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

    M = torch.zeros(n_out, n_in, dtype=torch.float32)
    scale = n_in / n_out
    for o in range(n_out):
        src = (o + 0.5) * scale - 0.5
        i0 = int(math.floor(src))
        t = src - i0
        ws = [c2(t + 1), c1(t), c1(1 - t), c2(2 - t)]
        for kk in range(4):
            idx = min(max(i0 - 1 + kk, 0), n_in - 1)
            M[o, idx] += ws[kk]
    return M


class Namespace(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _strip(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


# ---------------------------------------------------------------------------------------------------- towers
def prep_siglip(sd, heads, dtype, dev, patch=14, eps=1e-6, fp8=False, ln_fuse=False):
    """sd: HF SiglipVisionModel state dict (4.46 'vision_model.' prefix accepted).  siglip_encoder.py:71-78."""
    sd = {k.replace("vision_model.", ""): v for k, v in sd.items()}
    Wp = sd["embeddings.patch_embedding.weight"]
    D = Wp.shape[0]
    t = Namespace(kind="siglip", dim=D, heads=heads, head_dim=D // heads, patch=patch, eps=eps, has_cls=0,
                  act="gelu_tanh", final_ln=None, fp8=fp8_enabled(D, fp8))
    t.fused = ln_fusion_enabled(D, ln_fuse) and not t.fp8
    t.patch_lin = make_lin(Wp.reshape(D, -1), sd["embeddings.patch_embedding.bias"], dtype, dev)
    t.pos_table = sd["embeddings.position_embedding.weight"].detach().float().cpu()  # [P, D]
    t.layers = []
    i = 0
    while "encoder.layers.%d.layer_norm1.weight" % i in sd:
        p = "encoder.layers.%d." % i
        Lr = Namespace()
        Lr.ln1_g, Lr.ln1_b = vec32(sd[p + "layer_norm1.weight"], dev), vec32(sd[p + "layer_norm1.bias"], dev)
        # LayerNorm fusion (t.fused): LN1 of layers >= 1 and every LN2 are folded into the consuming GEMM: weight
        # W diag(gamma), bias beta W^T + b, c1 = row sums; layer 0's LN1 follows the patch embedding and stays a kernel
        f1 = dict(col_scale=sd[p + "layer_norm1.weight"], col_shift=sd[p + "layer_norm1.bias"]) if t.fused and i > 0 else None
        f2 = dict(col_scale=sd[p + "layer_norm2.weight"], col_shift=sd[p + "layer_norm2.bias"]) if t.fused else {}
        Lr.qkv = stack_lins([(sd[p + "self_attn.%s_proj.weight" % n], sd[p + "self_attn.%s_proj.bias" % n], f1)
                             for n in "qkv"], dtype, dev)
        Lr.qkv_c1 = fold_c1(Lr.qkv) if f1 else None
        Lr.out = make_lin(sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], dtype, dev)
        Lr.ln2_g, Lr.ln2_b = vec32(sd[p + "layer_norm2.weight"], dev), vec32(sd[p + "layer_norm2.bias"], dev)
        Lr.fc1 = make_lin(sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], dtype, dev, **f2)
        Lr.fc1_c1 = fold_c1(Lr.fc1) if f2 else None
        Lr.fc2 = make_lin(sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], dtype, dev)
        if t.fp8:
            Lr.qkv, Lr.fc1 = to_fp8(Lr.qkv, dev), to_fp8(Lr.fc1, dev)
        if t.fp8 >= 2:
            Lr.out, Lr.fc2 = to_fp8(Lr.out, dev), to_fp8(Lr.fc2, dev)
        t.layers.append(Lr)
        i += 1
    t.mlp = t.layers[0].fc1.n if t.layers else 0
    if t.fp8 >= 3 and not fp8_level3_ok(t):
        t.fp8 = 2
    return t


def prep_dino(sd, heads, dtype, dev, patch=14, eps=1e-6, fp8=False, ln_fuse=False):
    """sd: HF Dinov2Model state dict.  dino_encoder.py:109-120."""
    Wp = sd["embeddings.patch_embeddings.projection.weight"]
    D = Wp.shape[0]
    t = Namespace(kind="dino", dim=D, heads=heads, head_dim=D // heads, patch=patch, eps=eps, has_cls=1,
                  act="swiglu", fp8=fp8_enabled(D, fp8))
    t.fused = ln_fusion_enabled(D, ln_fuse) and not t.fp8
    t.patch_lin = make_lin(Wp.reshape(D, -1), sd["embeddings.patch_embeddings.projection.bias"], dtype, dev)
    t.pos_table = sd["embeddings.position_embeddings"].detach().float().cpu()[0]  # [1+n*n, D]
    t.cls = sd["embeddings.cls_token"].detach().float().cpu().flatten()
    t.layers = []
    i = 0
    while "encoder.layer.%d.norm1.weight" % i in sd:
        p = "encoder.layer.%d." % i
        Lr = Namespace()
        Lr.ln1_g, Lr.ln1_b = vec32(sd[p + "norm1.weight"], dev), vec32(sd[p + "norm1.bias"], dev)
        f1 = dict(col_scale=sd[p + "norm1.weight"], col_shift=sd[p + "norm1.bias"]) if t.fused and i > 0 else None
        f2 = dict(col_scale=sd[p + "norm2.weight"], col_shift=sd[p + "norm2.bias"]) if t.fused else {}
        Lr.qkv = stack_lins([(sd[p + "attention.attention.%s.weight" % n], sd[p + "attention.attention.%s.bias" % n],
                              f1) for n in ("query", "key", "value")], dtype, dev)
        Lr.qkv_c1 = fold_c1(Lr.qkv) if f1 else None
        Lr.out = make_lin(sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"], dtype, dev,
                          row_scale=sd[p + "layer_scale1.lambda1"])
        Lr.ln2_g, Lr.ln2_b = vec32(sd[p + "norm2.weight"], dev), vec32(sd[p + "norm2.bias"], dev)
        if p + "mlp.weights_in.weight" in sd:
            Win, bin_ = sd[p + "mlp.weights_in.weight"].float(), sd[p + "mlp.weights_in.bias"].float()
            hid = Win.shape[0] // 2
            hp = pad64(hid)
            Wi = torch.zeros(2 * hp, Win.shape[1], device=Win.device)
            bi = torch.zeros(2 * hp, device=Win.device)
            Wi[0:2 * hid:2], Wi[1:2 * hid:2] = Win[:hid], Win[hid:]
            bi[0:2 * hid:2], bi[1:2 * hid:2] = bin_[:hid], bin_[hid:]
            Lr.fc1 = make_lin(Wi, bi, dtype, dev, n_pad=2 * hp, **f2)
            _register_real_nk(Lr.fc1.w, 2 * hid, Win.shape[1])
            Lr.fc2 = make_lin(sd[p + "mlp.weights_out.weight"], sd[p + "mlp.weights_out.bias"], dtype, dev,
                              row_scale=sd[p + "layer_scale2.lambda1"])
            t.act = "swiglu"
        else:
            Lr.fc1 = make_lin(sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], dtype, dev, **f2)
            Lr.fc2 = make_lin(sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], dtype, dev,
                              row_scale=sd[p + "layer_scale2.lambda1"])
            t.act = "gelu_erf"
        Lr.fc1_c1 = fold_c1(Lr.fc1) if f2 else None
        if t.fp8:
            Lr.qkv, Lr.fc1 = to_fp8(Lr.qkv, dev), to_fp8(Lr.fc1, dev)
        if t.fp8 >= 2:
            Lr.out, Lr.fc2 = to_fp8(Lr.out, dev), to_fp8(Lr.fc2, dev)
        t.layers.append(Lr)
        i += 1
    t.final_ln = (vec32(sd["layernorm.weight"], dev), vec32(sd["layernorm.bias"], dev))
    if t.fp8 >= 3 and not fp8_level3_ok(t):
        t.fp8 = 2
    return t


def tower_pos(t, gh, gw, dev):
    """fp32 position rows for a gh x gw patch grid, padded to pad64(dim): [P (+1), ld].  Cached on the tower."""
    cache = t.setdefault("_pos_cache", {})
    key = (gh, gw)
    if key in cache:
        return cache[key]
    pos = t.pos_table
    if t.kind == "siglip":
        assert pos.shape[0] == gh * gw, "SigLIP learned positions: grid %dx%d != %d" % (gh, gw, pos.shape[0])
        table = mat32(pos, dev)
        cls_row = None
    else:
        n = int(round((pos.shape[0] - 1) ** 0.5))
        pp = pos[1:]
        if n * n != gh * gw or gh != gw:
            g = pp.reshape(n, n, -1)
            My, Mx = bicubic_matrix(n, gh), bicubic_matrix(n, gw)
            g = torch.einsum("oy,yxd->oxd", My, g)
            g = torch.einsum("px,oxd->opd", Mx, g)
            pp = g.reshape(gh * gw, -1)
        table = mat32(torch.cat([pos[:1], pp], 0), dev)
        cls_row = vec32(t.cls + pos[0], dev)
    cache[key] = (table, cls_row)
    return cache[key]


# ---------------------------------------------------------------------------------------------------- connector
def prep_connector(sd, cfg, dtype, dev):
    """sd: reference state dict with the leading 'model.' stripped.  cambrian_arch.py:65-109,140-150,469-484."""
    c = Namespace()
    H = cfg["hidden_size"]
    C = cfg["vision_hidden_size"]
    c.H, c.C = H, C
    c.aux = []
    for i in range(2):
        p = "mm_projector_aux_%d." % i
        c.aux.append(Namespace(fc1=make_lin(sd[p + "0.weight"], sd[p + "0.bias"], dtype, dev),
                               fc2=make_lin(sd[p + "2.weight"], sd[p + "2.bias"], dtype, dev),
                               ln_g=vec32(sd[p + "3.weight"], dev), ln_b=vec32(sd[p + "3.bias"], dev)))
    c.vision_query = sd["vision_query"].detach().float()[0]
    # SVA
    c.sva = []
    li = 0
    ones = torch.ones(C)
    while "vision_sampler_0.layers.%d.proj_in.weight" % li in sd:
        p = "vision_sampler_0.layers.%d." % li
        Lr = Namespace()
        Lr.proj_context = make_lin(sd[p + "proj_context.weight"], None, dtype, dev)
        Win = sd[p + "proj_in.weight"].float()
        Lr.proj_in_q = make_lin(Win[:, :C], None, dtype, dev)
        Lr.proj_in_c = make_lin(Win[:, C:], None, dtype, dev)
        Lr.q_ln = (vec32(sd[p + "cross_attn.q_proj.0.weight"], dev), vec32(sd[p + "cross_attn.q_proj.0.bias"], dev))
        Lr.q_proj = make_lin(sd[p + "cross_attn.q_proj.1.weight"], None, dtype, dev)
        Lr.kv = []
        Lr.pos = []
        for tw in range(2):
            gk, bk = sd[p + "cross_attn.k_proj_%d.0.weight" % tw], sd[p + "cross_attn.k_proj_%d.0.bias" % tw]
            gv, bv = sd[p + "cross_attn.v_proj_%d.0.weight" % tw], sd[p + "cross_attn.v_proj_%d.0.bias" % tw]
            Lr.kv.append(stack_lins([
                (sd[p + "cross_attn.k_proj_%d.1.weight" % tw], None, dict(col_scale=gk, col_shift=bk)),
                (sd[p + "cross_attn.v_proj_%d.1.weight" % tw], None, dict(col_scale=gv, col_shift=bv))],
                dtype, dev))
            Lr.pos.append(mat32(sd[p + "pos_embed_%d" % tw], dev))
        Lr.o_proj = make_lin(sd[p + "cross_attn.o_proj.weight"], None, dtype, dev)
        Lr.norm = (vec32(sd[p + "norm.weight"], dev), vec32(sd[p + "norm.bias"], dev))
        Lr.out1 = make_lin(sd[p + "proj_out.linear_1.weight"], None, dtype, dev)
        Lr.out2 = make_lin(sd[p + "proj_out.linear_2.weight"], None, dtype, dev)
        c.sva.append(Lr)
        li += 1
    c.ones_C = vec32(ones, dev)
    c.zeros_C = vec32(torch.zeros(C), dev)
    c.mm1 = make_lin(sd["mm_projector.0.weight"], sd["mm_projector.0.bias"], dtype, dev)
    c.mm2 = make_lin(sd["mm_projector.2.weight"], sd["mm_projector.2.bias"], dtype, dev)
    Hp = pad64(H)

    def row16(v):
        o = torch.zeros(1, Hp, device=v.device)
        o[0, :H] = v.detach().float()
        return o.to(dtype).to(dev)
    c.image_newline = row16(sd["image_newline"])
    c.frame_seg = row16(sd["frame_seg"])
    # Q-Former
    q = Namespace()
    p = "Qformer.bert."
    q.word = mat32(sd[p + "embeddings.word_embeddings.weight"], dev)
    q.pos = mat32(sd[p + "embeddings.position_embeddings.weight"], dev)
    q.emb_ln = (vec32(sd[p + "embeddings.LayerNorm.weight"], dev), vec32(sd[p + "embeddings.LayerNorm.bias"], dev))
    q.dim = q.word.shape[1] if False else sd[p + "embeddings.LayerNorm.weight"].numel()
    q.layers = []
    cross_parts = []
    li = 0
    while p + "encoder.layer.%d.attention.self.query.weight" % li in sd:
        lp = p + "encoder.layer.%d." % li
        Lr = Namespace()
        Lr.qkv = stack_lins([(sd[lp + "attention.self.%s.weight" % n], sd[lp + "attention.self.%s.bias" % n], None)
                             for n in ("query", "key", "value")], dtype, dev)
        Lr.attn_out = make_lin(sd[lp + "attention.output.dense.weight"], sd[lp + "attention.output.dense.bias"],
                               dtype, dev)
        Lr.attn_ln = (vec32(sd[lp + "attention.output.LayerNorm.weight"], dev),
                      vec32(sd[lp + "attention.output.LayerNorm.bias"], dev))
        Lr.cross = None
        if lp + "crossattention.self.query.weight" in sd:
            Lr.cross = Namespace(
                idx=len(cross_parts) // 2,
                q=make_lin(sd[lp + "crossattention.self.query.weight"], sd[lp + "crossattention.self.query.bias"],
                           dtype, dev),
                out=make_lin(sd[lp + "crossattention.output.dense.weight"],
                             sd[lp + "crossattention.output.dense.bias"], dtype, dev),
                ln=(vec32(sd[lp + "crossattention.output.LayerNorm.weight"], dev),
                    vec32(sd[lp + "crossattention.output.LayerNorm.bias"], dev)),
                q_tiled=None, out_tiled=None)   # fragment-major copies for tdc_qformer_xattn (pipeline._tile_cross_weights)
            cross_parts.append((sd[lp + "crossattention.self.key.weight"], sd[lp + "crossattention.self.key.bias"],
                                None))
            cross_parts.append((sd[lp + "crossattention.self.value.weight"],
                                sd[lp + "crossattention.self.value.bias"], None))
        for nm, a, b_ in (("ffn_q", "intermediate_query", "output_query"), ("ffn_t", "intermediate", "output")):
            Lr[nm] = Namespace(
                fc1=make_lin(sd[lp + a + ".dense.weight"], sd[lp + a + ".dense.bias"], dtype, dev),
                fc2=make_lin(sd[lp + b_ + ".dense.weight"], sd[lp + b_ + ".dense.bias"], dtype, dev),
                ln=(vec32(sd[lp + b_ + ".LayerNorm.weight"], dev), vec32(sd[lp + b_ + ".LayerNorm.bias"], dev)))
        q.layers.append(Lr)
        li += 1
    q.cross_kv = stack_lins(cross_parts, dtype, dev)  # [n_cross * 2 * dim, H]: (K_j | V_j) per cross layer j
    q.n_cross = len(cross_parts) // 2
    # the fused cross-attention block (tdc_qformer_xattn, bert-base width only) takes the same projections stacked per kind:
    # keys with their bias; values WITHOUT it (they are the A operand of the transposed GEMM V^T = Wv enc^T) and the value
    # biases as one fp32 vector that is added after the PV product
    q.cross_k = q.cross_v = q.cross_bv = None
    if q.dim == 768 and cross_parts:
        q.cross_k = stack_lins(cross_parts[0::2], dtype, dev)
        q.cross_v = stack_lins([(W, None, None) for (W, _, _) in cross_parts[1::2]], dtype, dev)
        q.cross_bv = torch.cat([b.detach().to(torch.float32).reshape(-1) for (_, b, _) in cross_parts[1::2]]).to(dev).contiguous()
    c.qformer = q
    c.query_proj = make_lin(sd["query_proj.weight"], sd["query_proj.bias"], dtype, dev)
    c.vision_proj = make_lin(sd["vision_proj.weight"], sd["vision_proj.bias"], dtype, dev)
    c.query_tokens = None
    if "query_tokens" in sd:
        qt = sd["query_tokens"].detach().reshape(-1, sd["query_tokens"].shape[-1]).to(torch.float32)
        buf = torch.zeros(qt.shape[0], pad64(qt.shape[1]), dtype=dtype, device=qt.device)
        buf[:, : qt.shape[1]] = qt.to(dtype)
        c.query_tokens = buf.to(dev)
    c.audio_proj = None
    if "audio_proj.weight" in sd:
        c.audio_proj = make_lin(sd["audio_proj.weight"], sd["audio_proj.bias"], dtype, dev)
    return c
