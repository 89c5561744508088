"""ORACLE — CPU restatement (torch fp32, eager) of TDC-Video's video-encoding hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file; the product
path (`tdc-video_amd/`) never does and fails loudly without its HIP library.

Parity status: the reference (Hoar012/TDC-Video) has no tests or golden vectors of its own and its ViT arithmetic
lives in third-party `transformers` (reference pins 4.46.0, requirements.txt:167; the build container has 5.15.0).
This restatement is therefore PINNED by fixtures generated from the imported reference in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz); tests/test_oracle_golden.py checks every function here
against them (<= 2e-5 abs in fp32, integers bit-exact).

All citations are file:line in /root/reference (tdc/...) or HF:<path>:line in transformers 5.15.0.
Weights are addressed by the reference's own state-dict names.
"""
import math

import torch
import torch.nn.functional as F

IMAGE_TOKEN_INDEX = -200  # tdc/constants.py
IGNORE_INDEX = -100


# ----------------------------------------------------------------------------------------------- small helpers
def _lin(x, W, name):
    b = W.get(name + ".bias")
    return F.linear(x, W[name + ".weight"], b)


def _ln(x, W, name, eps):
    return F.layer_norm(x, (x.shape[-1],), W[name + ".weight"], W[name + ".bias"], eps)


def sub(W, prefix):
    """View of a state dict restricted to `prefix` (prefix stripped)."""
    n = len(prefix)
    return {k[n:]: v for k, v in W.items() if k.startswith(prefix)}


def bilinear_matrix(n_in, n_out):
    """Rows of F.interpolate(mode='bilinear', align_corners=False) along one axis as an [n_out, n_in] matrix.
    src = (dst + 0.5) * n_in / n_out - 0.5, clamped at 0; weights (1-frac, frac) on (i0, min(i0+1, n_in-1)).
    Used at siglip_encoder.py:56-61 / dino_encoder.py:94-99 (27x27 -> 24x24)."""
    M = torch.zeros(n_out, n_in, dtype=torch.float32)
    scale = n_in / n_out
    for o in range(n_out):
        src = (o + 0.5) * scale - 0.5
        if src < 0:
            src = 0.0
        i0 = int(math.floor(src))
        i1 = min(i0 + 1, n_in - 1)
        fr = src - i0
        M[o, i0] += 1.0 - fr
        M[o, i1] += fr
    return M


def bicubic_matrix(n_in, n_out, A=-0.75):
    """F.interpolate(mode='bicubic', align_corners=False) along one axis as [n_out, n_in] (cubic convolution,
    A=-0.75, border indices clamped).  Used by HF:models/dinov2/modeling_dinov2.py:79-88 to resample the
    37x37 position table to the 27x27 patch grid."""
    def c1(x):  # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):  # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

    M = torch.zeros(n_out, n_in, dtype=torch.float32)
    scale = n_in / n_out
    for o in range(n_out):
        src = (o + 0.5) * scale - 0.5
        i0 = int(math.floor(src))
        t = src - i0
        ws = [c2(t + 1), c1(t), c1(1 - t), c2(2 - t)]
        for k in range(4):
            idx = min(max(i0 - 1 + k, 0), n_in - 1)
            M[o, idx] += ws[k]
    return M


def resize_tokens(x, n_out, matfn=bilinear_matrix):
    """x [B, n*n, D] on an n x n grid -> [B, n_out*n_out, D] (separable resample)."""
    B, P, D = x.shape
    n = int(round(P ** 0.5))
    if n == n_out:
        return x
    M = matfn(n, n_out)
    g = x.reshape(B, n, n, D).to(torch.float32)
    g = torch.einsum("oy,byxd->boxd", M, g)
    g = torch.einsum("px,boxd->bopd", M, g)
    return g.reshape(B, n_out * n_out, D).to(x.dtype)


def patchify(px, p):
    """[B,3,H,W] -> [B, gh*gw, 3*p*p] in conv-weight order (c, ky, kx); stride-p 'valid' conv as a GEMM."""
    B, C, H, Wd = px.shape
    gh, gw = H // p, Wd // p
    x = px[:, :, : gh * p, : gw * p].reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, gh * gw, C * p * p), gh, gw


def mha(q, k, v, heads, scale):
    """softmax(q k^T * scale) v over [B, S, D] tensors split in `heads`."""
    B, Sq, D = q.shape
    Sk = k.shape[1]
    d = D // heads
    q = q.reshape(B, Sq, heads, d).transpose(1, 2)
    k = k.reshape(B, Sk, heads, d).transpose(1, 2)
    v = v.reshape(B, Sk, heads, d).transpose(1, 2)
    a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * scale, dim=-1)
    return torch.matmul(a, v).transpose(1, 2).reshape(B, Sq, D)


# ----------------------------------------------------------------------------------------------- a3  SigLIP tower
def siglip_tower(px, W, heads, patch=14, interp_tokens=576, eps=1e-6):
    """SiglipVisionTower._forward (tdc/multimodal_encoder/siglip_encoder.py:71-78): HF SiglipVisionModel
    embeddings + encoder, taking hidden_states[-1] (last block output, BEFORE post_layernorm, no pooling head),
    then bilinear 27x27 -> 24x24 (:43-69).  HF:models/siglip/modeling_siglip.py:116-186 (embeddings),
    :250-308 (attention, scale=d^-0.5), :310-357 (tanh-GELU MLP, pre-LN residual block)."""
    W = {k.replace("vision_model.", ""): v for k, v in W.items()}
    x, gh, gw = patchify(px, patch)
    D = W["embeddings.patch_embedding.weight"].shape[0]
    x = F.linear(x, W["embeddings.patch_embedding.weight"].reshape(D, -1), W["embeddings.patch_embedding.bias"])
    x = x + W["embeddings.position_embedding.weight"][None]
    L = 0
    while "encoder.layers.%d.layer_norm1.weight" % L in W:
        L += 1
    d = D // heads
    for i in range(L):
        p = "encoder.layers.%d." % i
        h = _ln(x, W, p + "layer_norm1", eps)
        a = mha(_lin(h, W, p + "self_attn.q_proj"), _lin(h, W, p + "self_attn.k_proj"),
                _lin(h, W, p + "self_attn.v_proj"), heads, d ** -0.5)
        x = x + _lin(a, W, p + "self_attn.out_proj")
        h = _ln(x, W, p + "layer_norm2", eps)
        h = F.gelu(_lin(h, W, p + "mlp.fc1"), approximate="tanh")
        x = x + _lin(h, W, p + "mlp.fc2")
    pre = x
    return resize_tokens(x, int(round(interp_tokens ** 0.5))), pre


# ----------------------------------------------------------------------------------------------- a4  DINOv2 tower
def dino_tower(px, W, heads, patch=14, interp_tokens=576, eps=1e-6):
    """DinoVisionTower._forward (tdc/multimodal_encoder/dino_encoder.py:109-120): HF Dinov2Model -> last_hidden_state
    (after the final LayerNorm) -> drop cls (feature_select 'patch', :66-79) -> bilinear to 24x24 (:81-107).
    HF:models/dinov2/modeling_dinov2.py:38-116 (cls + bicubic-resampled pos), :182-236 (attention),
    :272-314 (LayerScale, SwiGLU: silu(x1)*x2), :342-380 (block), :433-470 (final layernorm)."""
    x, gh, gw = patchify(px, patch)
    D = W["embeddings.patch_embeddings.projection.weight"].shape[0]
    x = F.linear(x, W["embeddings.patch_embeddings.projection.weight"].reshape(D, -1),
                 W["embeddings.patch_embeddings.projection.bias"])
    B = x.shape[0]
    x = torch.cat([W["embeddings.cls_token"].expand(B, -1, -1), x], dim=1)
    pos = W["embeddings.position_embeddings"]  # [1, 1+n*n, D]
    npos = pos.shape[1] - 1
    if npos != gh * gw:
        pp = resize_tokens(pos[:, 1:], gh, bicubic_matrix)
        pos = torch.cat([pos[:, :1], pp], dim=1)
    x = x + pos
    L = 0
    while "encoder.layer.%d.norm1.weight" % L in W:
        L += 1
    d = D // heads
    for i in range(L):
        p = "encoder.layer.%d." % i
        h = _ln(x, W, p + "norm1", eps)
        a = mha(_lin(h, W, p + "attention.attention.query"), _lin(h, W, p + "attention.attention.key"),
                _lin(h, W, p + "attention.attention.value"), heads, d ** -0.5)
        a = _lin(a, W, p + "attention.output.dense") * W[p + "layer_scale1.lambda1"]
        x = x + a
        h = _ln(x, W, p + "norm2", eps)
        if (p + "mlp.weights_in.weight") in W:
            h = _lin(h, W, p + "mlp.weights_in")
            x1, x2 = h.chunk(2, dim=-1)
            h = _lin(F.silu(x1) * x2, W, p + "mlp.weights_out")
        else:
            h = _lin(F.gelu(_lin(h, W, p + "mlp.fc1")), W, p + "mlp.fc2")
        x = x + h * W[p + "layer_scale2.lambda1"]
    x = _ln(x, W, "layernorm", eps)
    pre = x
    return resize_tokens(x[:, 1:], int(round(interp_tokens ** 0.5))), pre


# ----------------------------------------------------------------------------------------------- a1  frame cap
def get_max_num_frames(cur_input_ids, cfg):
    """tdc/cambrian_arch.py:748-780."""
    pad_id = 128002 if "llama" in cfg.get("model_type", "") else 151643
    pos = torch.where(cur_input_ids == pad_id)[-1]
    text_len = int(pos[0]) if len(pos) > 0 else len(cur_input_ids)
    K = cfg.get("context_token_num", 16)
    if not cfg.get("audio_input", False):
        tpf = (144 + K * 7) // 8
    else:
        tpf = (144 + 50 + K * 7) // 8
    if not cfg.get("add_static", True):
        tpf = 16
    return max(1, (cfg["tokenizer_model_max_length"] - text_len - cfg.get("inference_max_length", 16)) // tpf)


def frame_cap_indices(T, max_num_frames, frame_cap=224):
    """tdc/cambrian_arch.py:907-935: uniform subsampling to min(budget, 224) frames; returns (indices, sample_indices)."""
    mx = min(max_num_frames, frame_cap)
    if T > mx:
        interval = T / float(mx)
        idx = [int(interval * i) for i in range(mx)]
        samp = torch.zeros(T, dtype=torch.int16)
        samp[idx] = 1
    else:
        idx = list(range(T))
        samp = torch.ones(T, dtype=torch.int16)
    return idx, samp


# ----------------------------------------------------------------------------------------------- a5  segmentation
def adjacent_cosine(feat, window=64):
    """cos-sim of consecutive frames on flattened features, 64-pair windows (tdc/cambrian_arch.py:832-842)."""
    q = feat.flatten(1, 2)
    prev, nxt = q[:-1], q[1:]
    sims = []
    for s in range(0, len(prev), window):
        sims.append(F.cosine_similarity(prev[s:s + window], nxt[s:s + window], dim=1))
    return torch.cat(sims)


def select_segments(sims, max_num_segments=24):
    """argsort ascending, keep the `max_num_segments` lowest, sort (tdc/cambrian_arch.py:849).
    Ties: the restatement uses a STABLE sort (lowest index first); torch.argsort's default is unstable, so
    fixtures use well-separated similarities."""
    order = torch.argsort(sims, stable=True)[:max_num_segments]
    return torch.sort(order)[0]


def adapt_segment(dino_feat, max_num_segments=24, seg_frame_cap=224):
    """tdc/cambrian_arch.py:783-861 for one video: returns (selected_frame_indices, seg_indices)."""
    T = len(dino_feat)
    if T <= max_num_segments + 1:
        return torch.arange(T), torch.arange(T)
    if T > seg_frame_cap:
        interval = T / float(seg_frame_cap)
        idx = [int(interval * i) for i in range(seg_frame_cap)]
    else:
        idx = list(range(T))
    sims = adjacent_cosine(dino_feat[idx])
    return torch.tensor(idx), select_segments(sims, max_num_segments)


# ----------------------------------------------------------------------------------------------- a6  aux projectors
def mm_projector_aux(x, W, i):
    """Linear -> GELU(erf) -> Linear -> LayerNorm(1e-5) (tdc/cambrian_arch.py:80-90, applied :1002-1013)."""
    p = "mm_projector_aux_%d." % i
    h = F.gelu(_lin(x, W, p + "0"))
    return _ln(_lin(h, W, p + "2"), W, p + "3", 1e-5)


# ----------------------------------------------------------------------------------------------- a7  geometry
def unpad_bounds(cur_h, cur_w, image_size):
    """Index range kept by unpad_image (tdc/cambrian_arch.py:512-544).  NOTE `image_size` is unpacked as
    (width, height) although callers pass (height, width) (main.py:36) - reproduced on purpose.
    Returns (row0, row1, col0, col1)."""
    ow, oh = image_size
    if ow / oh > cur_w / cur_h:
        new_h = int(oh * (cur_w / ow))
        pad = (cur_h - new_h) // 2
        return pad, cur_h - pad, 0, cur_w
    new_w = int(ow * (cur_h / oh))
    pad = (cur_w - new_w) // 2
    return 0, cur_h, pad, cur_w - pad


def window_masks(side, reduce, image_size):
    """Boolean kv masks [side*side, reduce*reduce] for one frame (tdc/cambrian_arch.py:487-509, :619-669):
    padding rows/cols of the aux grid are masked, then windows whose mask is all False are forced True."""
    n = side * reduce
    mask = torch.ones(n, n, dtype=torch.bool)
    ow, oh = image_size
    if ow / oh > 1.0:  # cur_w / cur_h == 1
        new_h = int(oh * (n / ow))
        pad = (n - new_h) // 2
        if pad > 0:
            mask[:pad, :] = False
            mask[-pad:, :] = False
    else:
        new_w = int(ow * (n / oh))
        pad = (n - new_w) // 2
        if pad > 0:
            mask[:, :pad] = False
            mask[:, -pad:] = False
    m = mask.view(side, reduce, side, reduce).permute(0, 2, 1, 3).reshape(side * side, reduce * reduce).clone()
    m[m.sum(-1) == 0] = True
    return m


def rearrange_windows(feat, side):
    """[T, (side*r)^2, C] -> [T*side*side, r*r, C] (tdc/cambrian_arch.py:624-645)."""
    T, P, C = feat.shape
    n = int(round(P ** 0.5))
    r = n // side
    return feat.view(T, side, r, side, r, C).permute(0, 1, 3, 2, 4, 5).reshape(T * side * side, r * r, C)


# ----------------------------------------------------------------------------------------------- a8  SVA
def sva_layer(queries, ctx, kv_list, mask_list, W, p, heads=16):
    """VisionCrossAttentionLayer.forward (tdc/vision_sampler.py:341-401) with MultiKVCrossAttention (:215-291)."""
    residual = queries
    c = F.linear(ctx, W[p + "proj_context.weight"])
    q = F.linear(torch.cat([queries, c], -1), W[p + "proj_in.weight"])
    ks, vs = [], []
    for i, kv in enumerate(kv_list):
        kvp = kv + W[p + "pos_embed_%d" % i][None]
        kn = F.layer_norm(kvp, (kvp.shape[-1],), W[p + "cross_attn.k_proj_%d.0.weight" % i],
                          W[p + "cross_attn.k_proj_%d.0.bias" % i], 1e-5)
        vn = F.layer_norm(kvp, (kvp.shape[-1],), W[p + "cross_attn.v_proj_%d.0.weight" % i],
                          W[p + "cross_attn.v_proj_%d.0.bias" % i], 1e-5)
        ks.append(F.linear(kn, W[p + "cross_attn.k_proj_%d.1.weight" % i]))
        vs.append(F.linear(vn, W[p + "cross_attn.v_proj_%d.1.weight" % i]))
    k = torch.cat(ks, 1)
    v = torch.cat(vs, 1)
    mask = torch.cat(mask_list, -1)  # [B, kv]
    qn = F.layer_norm(q, (q.shape[-1],), W[p + "cross_attn.q_proj.0.weight"], W[p + "cross_attn.q_proj.0.bias"], 1e-5)
    qs = F.linear(qn, W[p + "cross_attn.q_proj.1.weight"])
    B, _, D = qs.shape
    d = D // heads
    qh = qs.view(B, 1, heads, d).transpose(1, 2)
    kh = k.view(B, -1, heads, d).transpose(1, 2)
    vh = v.view(B, -1, heads, d).transpose(1, 2)
    s = torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(d)
    s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    a = torch.matmul(torch.softmax(s, -1), vh).transpose(1, 2).reshape(B, 1, D)
    q = q + F.linear(a, W[p + "cross_attn.o_proj.weight"])
    q = F.layer_norm(q, (D,), W[p + "norm.weight"], W[p + "norm.bias"], 1e-5)
    q = F.linear(F.gelu(F.linear(q, W[p + "proj_out.linear_1.weight"])), W[p + "proj_out.linear_2.weight"])
    return q + residual


def sva(aux_list, vision_query, image_sizes, W, side, prefix="vision_sampler_0."):
    """tdc/cambrian_arch.py:1017-1053 + VisionTokenSampler.forward (tdc/vision_sampler.py:561-566).
    aux_list: per-tower [T, P, C]; returns [T, side*side, C]."""
    T = aux_list[0].shape[0]
    nq = side * side
    ctx = aux_list[0].mean(1).view(T, 1, 1, -1).expand(-1, nq, 1, -1).flatten(0, 1)
    q = vision_query.view(1, 1, 1, -1).expand(T, nq, -1, -1).flatten(0, 1)
    kv = [rearrange_windows(a, side) for a in aux_list]
    masks = []
    for a in aux_list:
        r = int(round(a.shape[1] ** 0.5)) // side
        masks.append(torch.cat([window_masks(side, r, image_sizes[t]) for t in range(T)], 0))
    L = 0
    while (prefix + "layers.%d.proj_in.weight" % L) in W:
        L += 1
    for i in range(L):
        q = sva_layer(q, ctx, kv, masks, W, prefix + "layers.%d." % i)
    return q.view(T, nq, -1), masks


# ----------------------------------------------------------------------------------------------- a9/a10
def mm_projector(x, W):
    """Linear -> GELU(erf) -> Linear (tdc/cambrian_arch.py:65-69, applied :1149-1150)."""
    return _lin(F.gelu(_lin(x, W, "mm_projector.0")), W, "mm_projector.2")


def unpad_newline(feat, image_sizes, newline):
    """tdc/cambrian_arch.py:1176-1293: per frame view side x side, unpad, append the newline column, flatten.
    Returns (list of [n_t, H], final_size list).  The try/except at :1205-1214 never triggers for valid sizes."""
    T, nq, H = feat.shape
    side = int(round(nq ** 0.5))
    out, sizes = [], []
    for t in range(T):
        r0, r1, c0, c1 = unpad_bounds(side, side, image_sizes[t])
        cur = feat[t].view(side, side, H)[r0:r1, c0:c1]
        ch, cw = cur.shape[:2]
        sizes.append((ch, cw))
        cur = torch.cat([cur, newline.view(1, 1, -1).expand(ch, 1, -1)], dim=1)
        out.append(cur.reshape(ch * (cw + 1), H))
    return out, sizes


# ----------------------------------------------------------------------------------------------- a12-a18 Q-Former
def adaptive_avg_pool_tokens(x, K):
    """adaptive_avg_pool1d over the token axis: window i = [floor(i*N/K), ceil((i+1)*N/K)) (cambrian_arch.py:1634-1637)."""
    N = x.shape[0]
    rows = []
    for i in range(K):
        s = (i * N) // K
        e = -((-(i + 1) * N) // K)
        rows.append(x[s:e].mean(0))
    return torch.stack(rows)


def qformer_bert(query_embeds, enc, prompt_ids, W, heads, p="Qformer.bert.", eps=1e-12, cross_freq=2):
    """BertModel.forward (tdc/Qformer.py:804-965) for this path: masks are all-zero, no cache.
    query_embeds [L,K,Dq], enc [L,N,H], prompt_ids [Lt] or None -> last_hidden_state [L, K+Lt, Dq]."""
    L, K, D = query_embeds.shape
    # BertEmbeddings (:78-108)
    if prompt_ids is not None:
        Lt = len(prompt_ids)
        te = W[p + "embeddings.word_embeddings.weight"][prompt_ids] + \
            W[p + "embeddings.position_embeddings.weight"][:Lt]
        x = torch.cat([query_embeds, te[None].expand(L, -1, -1)], dim=1)
    else:
        x = query_embeds
    x = _ln(x, W, p + "embeddings.LayerNorm", eps)
    nl = 0
    while (p + "encoder.layer.%d.attention.self.query.weight" % nl) in W:
        nl += 1
    d = D // heads
    for i in range(nl):
        lp = p + "encoder.layer.%d." % i
        # self attention over all S rows (:417-423, :169-275, :285-289)
        a = mha(_lin(x, W, lp + "attention.self.query"), _lin(x, W, lp + "attention.self.key"),
                _lin(x, W, lp + "attention.self.value"), heads, 1.0 / math.sqrt(d))
        x = _ln(_lin(a, W, lp + "attention.output.dense") + x, W, lp + "attention.output.LayerNorm", eps)
        qx = x[:, :K]
        if i % cross_freq == 0:  # (:386-393, :432-447)
            a = mha(_lin(qx, W, lp + "crossattention.self.query"), _lin(enc, W, lp + "crossattention.self.key"),
                    _lin(enc, W, lp + "crossattention.self.value"), heads, 1.0 / math.sqrt(d))
            qx = _ln(_lin(a, W, lp + "crossattention.output.dense") + qx, W, lp + "crossattention.output.LayerNorm", eps)
        # dual FFN (:449-462, :476-484)
        h = F.gelu(_lin(qx, W, lp + "intermediate_query.dense"))
        qo = _ln(_lin(h, W, lp + "output_query.dense") + qx, W, lp + "output_query.LayerNorm", eps)
        if x.shape[1] > K:
            tx = x[:, K:]
            h = F.gelu(_lin(tx, W, lp + "intermediate.dense"))
            to = _ln(_lin(h, W, lp + "output.dense") + tx, W, lp + "output.LayerNorm", eps)
            x = torch.cat([qo, to], dim=1)
        else:
            x = qo
    return x


def compress_chunk(chunk, prompt_ids, W, K, heads, audio_chunk=None, add_static=True, query_type="Avg_pool"):
    """One <=8-frame chunk (tdc/cambrian_arch.py:1608-1667): key = chunk[0]; returns compressed [L,K,H] (L2-normalised)
    and the chunk features actually used (with audio tokens appended when given, :1611-1614).  add_static=False: the
    key frame itself is compressed too (:1625-1628); query_type='learned': query_tokens instead of the pooled key
    frame (:1633-1640)."""
    key = chunk[0]  # visual-only key frame (:1609) - pooled BEFORE the audio concat
    if audio_chunk is not None:
        chunk = torch.cat([chunk, _lin(audio_chunk, W, "audio_proj")], dim=1)
    vin = chunk[1:] if add_static else chunk
    L = vin.shape[0]
    if query_type == "learned":
        q = W["query_tokens"][0]
    else:
        q = _lin(adaptive_avg_pool_tokens(key, K), W, "query_proj")  # [K, Dq]
    last = qformer_bert(q[None].expand(L, -1, -1), vin, prompt_ids, W, heads)
    comp = F.normalize(_lin(last[:, :K], W, "vision_proj"), dim=-1)
    return comp, chunk


def audio_tokens(beats_windows, sample_indices, n_frames, dist=10):
    """a20 host part (tdc/cambrian_arch.py:1552-1598): BEATs features of consecutive 10-s windows ([1, n_w, 768],
    50 tokens per second) -> [n_frames, 50, 768].  A second whose frame was dropped by the frame cap (sample_indices
    == 0) is average-pooled together with the preceding kept second; short slices are adaptively pooled to 50 tokens;
    the tail is zero padded.  `sample_indices` is the 0/1 vector of a1 (all ones when no cap)."""
    audio_embeds, seg = [], []
    pool = lambda x: F.adaptive_avg_pool2d(x, (50, x.shape[-1]))
    for w, emb in enumerate(beats_windows):
        k = w * dist
        window = sample_indices[k:k + dist]
        sample_len = len(window)
        for idx, ind in enumerate(window):
            token = emb[:, idx * 50:(idx + 1) * 50, :]
            if token.shape[1] == 0:
                continue
            if token.shape[1] != 50:
                token = pool(token)
            if ind == 1:
                if seg:
                    audio_embeds.append(pool(torch.cat(seg, dim=1)))
                    seg = []
                seg.append(token)
                if idx + 1 < sample_len and sample_indices[k + idx + 1] == 1:
                    audio_embeds.append(token)
                    seg = []
            elif ind == 0:
                seg.append(token)
    if seg:
        audio_embeds.append(pool(torch.cat(seg, dim=1)))
    a = torch.cat(audio_embeds).flatten(0, 1)
    pad = n_frames * 50 - a.shape[0]
    a = F.pad(a, (0, 0, 0, pad))
    return a.reshape(-1, 50, a.shape[-1])


def chunk_table(T, seg_indices):
    """Segments -> <=8-frame chunks (tdc/cambrian_arch.py:1541-1545, :1603-1608): list of (start, end)."""
    split_points = [0] + [int(s) + 1 for s in seg_indices] + [T]
    chunks = []
    for a, b in zip(split_points[:-1], split_points[1:]):
        for s in range(a, b, 8):
            chunks.append((s, min(s + 8, b)))
    return chunks


def tdc_compress(frames, seg_indices, prompt_ids, W, K, heads, max_visual_len, audio=None, add_static=True,
                 query_type="Avg_pool"):
    """S10 (tdc/cambrian_arch.py:1520-1709): frames [T,N,H] -> emitted visual tokens [n,H].
    add_sep=True (hard-wired at :1512); add_static / query_type per config (:1509-1511); text_input per prompt_ids
    (None = no text)."""
    T = frames.shape[0]
    fseg = W["frame_seg"][None]
    out = []
    for (s, e) in chunk_table(T, seg_indices):
        chunk = frames[s:e]
        ach = audio[s:e] if audio is not None else None
        if add_static and e - s == 1:
            c0 = chunk[0]
            if ach is not None:
                c0 = torch.cat([c0, _lin(ach[0], W, "audio_proj")], dim=0)
            out.append(torch.cat([c0, fseg]))
            continue
        comp, chunk_full = compress_chunk(chunk, prompt_ids, W, K, heads, ach, add_static, query_type)
        L = comp.shape[0]
        ctx = torch.cat([comp, fseg[None].expand(L, -1, -1)], dim=1).flatten(0, 1)
        out.append(torch.cat([torch.cat([chunk_full[0], fseg]), ctx], dim=0) if add_static else ctx)
    total = sum(x.shape[0] for x in out)
    if total > max_visual_len:
        rm = math.ceil((total - max_visual_len) / len(out))
        out = [x[:-rm] for x in out]
    return torch.cat(out, dim=0)[:max_visual_len]


# ----------------------------------------------------------------------------------------------- top level
def encode_video(W, cfg, px_siglip, px_dino, image_size, input_ids, prompt_ids, audio=None, frame_cap=224,
                 beats_windows=None):
    """prepare_inputs_labels_for_multimodal (tdc/cambrian_arch.py:864-1844) for ONE video sample
    (batch 1, one <image> token), returning inputs_embeds [1,S,H] plus every intermediate.
    W: reference-named state dict with the leading 'model.' stripped; cfg: dict of config keys."""
    r = {}
    T0 = px_siglip.shape[0]
    ids = input_ids[0]
    idx, samp = frame_cap_indices(T0, get_max_num_frames(ids, cfg), frame_cap)            # a1
    r["frame_indices"], r["sample_indices"] = idx, samp
    px_s, px_d = px_siglip[idx], px_dino[idx]
    tok = cfg.get("mm_vision_tower_aux_token_len_list", [576, 576])
    Wd = sub(W, "vision_tower_aux_list.1.vision_tower.")
    Ws = sub(W, "vision_tower_aux_list.0.vision_tower.")
    dino = torch.cat([dino_tower(px_d[s:s + 64], Wd, cfg["dino_heads"], interp_tokens=tok[1])[0]
                      for s in range(0, len(idx), 64)])                                   # a2 + a4
    sel, seg = adapt_segment(dino, cfg.get("max_num_segments", 24), frame_cap)            # a5
    r["selected"], r["seg_indices"] = sel, seg
    dino = dino[sel]
    px_s = px_s[sel]
    sig = torch.cat([siglip_tower(px_s[s:s + 64], Ws, cfg["siglip_heads"], interp_tokens=tok[0])[0]
                     for s in range(0, len(sel), 64)])                                    # a3
    r["siglip_feat"], r["dino_feat"] = sig, dino
    T = len(sel)
    sizes = [tuple(image_size)] * T
    aux = [mm_projector_aux(sig, W, 0), mm_projector_aux(dino, W, 1)]                     # a6
    r["aux0"], r["aux1"] = aux
    side = int(round(cfg["query_num_list"][0] ** 0.5))
    q, _ = sva(aux, W["vision_query"][0], sizes, W, side)                                 # a7 + a8
    r["sva"] = q
    feat = mm_projector(q, W)                                                             # a9
    r["mm_proj"] = feat
    frames, final_size = unpad_newline(feat, sizes, W["image_newline"])                   # a10
    r["final_size"] = final_size
    frames = torch.stack(frames)  # one video: all frames share image_size
    # a21: text split / embed
    pos = int(torch.where(ids == IMAGE_TOKEN_INDEX)[0][0])
    pre, post = ids[:pos], ids[pos + 1:]
    emb = W["embed_tokens_fn"]
    text_len = len(pre) + len(post)
    max_visual_len = cfg["tokenizer_model_max_length"] - cfg.get("inference_max_length", 16) - text_len
    if beats_windows is not None:
        audio = audio_tokens(beats_windows, samp, T)                                                  # a20
        r["audio_tokens"] = audio
    vis = tdc_compress(frames, seg, prompt_ids if cfg.get("text_input", True) else None, W,
                       cfg.get("context_token_num", 16), cfg["qformer_heads"], max_visual_len, audio,
                       cfg.get("add_static", True), cfg.get("query_type", "Avg_pool"))                 # a11-a20
    r["visual_tokens"] = vis
    x = torch.cat([emb(pre), vis, emb(post)], dim=0)[: cfg["tokenizer_model_max_length"]]
    r["inputs_embeds"] = x[None]
    return r
