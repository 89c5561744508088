"""BEATs audio encoder on the HIP kernels: the `extract_features(wav, padding_mask, feature_only=True)`
call of tdc/cambrian_arch.py:1552-1560, i.e. tdc/audio_models/beats/BEATs.py:131-178 + backbone.py:98-274.

Device sequence for B equal-length windows (L = (frames//16)*8 tokens each, C = 768):
  tdc_fbank (fbank + normalisation, written as patch-conv im2col) -> GEMM (patch_embedding) -> LN -> GEMMs
  (post_extract_proj, once channel-last and once group-major) -> conv positional embedding as 16 GEMMs whose A operand
  is an OVERLAPPING-row view of the zero-padded group-major sequence (row stride = channels per group, K = kernel x
  channels: the im2col matrix is never materialised; bias + GELU in the epilogue, residual in the LN) -> LN -> 12 post-LN
  deep-norm layers: fused QKV GEMM, tdc_relpos_gate, tdc_attention with the gated relative position bias, out_proj
  (+ residual, fp32 out), LN, fc1 + GELU, fc2 (+ residual, fp32 out), LN.
deep-norm: LN(alpha*x + f(x)) == LN_{eps/alpha^2}(x + f(x)/alpha) exactly, so 1/alpha is folded into out_proj / fc2 and
eps into the LayerNorm (no scaled-residual epilogue needed).

Weights: the state dict of the reference's BEATs module (`beats_ckpt['model']`, audio_encoder.py:61-65), either
weight_norm spelling of pos_conv.  No CPU fallback: everything after the host-side table construction runs through
libtdc_hip.so."""
import math

import torch

from . import lib as L
from . import ops
from .weights import make_lin, pad64, stack_lins

BEATS_ITER3_CFG = dict(   # cfg stored in BEATs_iter3_plus_AS2M_finetuned_on_AS2M_cpt2.pt (cambrian_arch.py:454)
    input_patch_size=16, embed_dim=512, conv_bias=False, encoder_layers=12, encoder_embed_dim=768,
    encoder_ffn_embed_dim=3072, encoder_attention_heads=12, activation_fn="gelu", layer_norm_first=False,
    deep_norm=True, conv_pos=128, conv_pos_groups=16, relative_position_embedding=True, num_buckets=320,
    max_distance=800, gru_rel_pos=True)

FBANK_MEAN, FBANK_STD = 15.41663, 6.55582   # BEATs.py:118-119
SAMPLE_RATE = 16000


def fbank_tables(dev):
    """host-built constants of torchaudio.compliance.kaldi.fbank(num_mel_bins=128, 16 kHz, 25/10 ms): povey window,
    FFT twiddles, mel filter bank (+ the non-zero span of every mel row)."""
    window = torch.hann_window(400, periodic=False).pow(0.85)
    k = torch.arange(256, dtype=torch.float64)
    tw = torch.stack([torch.cos(2 * math.pi * k / 512), torch.sin(2 * math.pi * k / 512)], 1).float().contiguous()
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    nbins, nfft, width = 128, 256, 16000.0 / 512
    lo, hi = mel(20.0), mel(8000.0)
    delta = (hi - lo) / (nbins + 1)
    b = torch.arange(nbins).unsqueeze(1)
    left, center, right = lo + b * delta, lo + (b + 1.0) * delta, lo + (b + 2.0) * delta
    m = (1127.0 * (1.0 + width * torch.arange(nfft) / 700.0).log()).unsqueeze(0)
    banks = torch.max(torch.zeros(1), torch.min((m - left) / (center - left), (right - m) / (right - center)))
    banks = torch.nn.functional.pad(banks, (0, 1)).contiguous()          # [128, 257]
    rng = torch.zeros(nbins, 2, dtype=torch.int32)
    for i in range(nbins):
        nz = torch.nonzero(banks[i] > 0).flatten()
        if nz.numel():
            rng[i, 0], rng[i, 1] = int(nz[0]), int(nz[-1]) + 1
    return window.to(dev), tw.to(dev), banks.to(dev), rng.to(dev)


def relative_position_bucket(rel, num_buckets, max_distance):
    """backbone.py:391-416 on the host (same torch CPU ops as the reference, which also builds the table on the CPU)."""
    nb = num_buckets // 2
    out = (rel > 0).to(torch.long) * nb
    rel = torch.abs(rel)
    max_exact = nb // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(is_small, rel, large)


def _pos_conv_weight(sd, pre="encoder.pos_conv.0."):
    if pre + "weight" in sd:
        return sd[pre + "weight"].float()
    if pre + "weight_g" in sd:
        g, v = sd[pre + "weight_g"].float(), sd[pre + "weight_v"].float()
    else:
        g, v = sd[pre + "parametrizations.weight.original0"].float(), sd[pre + "parametrizations.weight.original1"].float()
    return v * (g / v.norm(dim=(0, 1), keepdim=True))


class _Layer:
    pass


class BeatsEncoder:
    def __init__(self, sd, cfg=None, dtype=torch.float16, device="cuda"):
        cfg = dict(BEATS_ITER3_CFG if cfg is None else cfg)
        if cfg.get("layer_norm_first", False) or cfg.get("activation_fn", "gelu") != "gelu":
            raise NotImplementedError("only the post-LN / GELU BEATs configuration of the released checkpoints")
        if cfg["input_patch_size"] != 16:
            raise NotImplementedError("patch size 16 (the fbank kernel writes 16x16 patches)")
        self.cfg, self.dtype, self.dev = cfg, dtype, torch.device(device)
        dev = self.dev
        sd = {k: v.detach() for k, v in sd.items()}
        C = self.C = cfg["encoder_embed_dim"]
        E = self.E = cfg["embed_dim"]
        self.heads = cfg["encoder_attention_heads"]
        self.hd = C // self.heads
        self.G, self.kk = cfg["conv_pos_groups"], cfg["conv_pos"]
        self.cg = C // self.G
        assert self.hd % 8 == 0 and self.hd <= 64 and self.cg % 8 == 0 and (self.kk * self.cg) % 64 == 0
        nl = cfg["encoder_layers"]
        self.alpha = math.pow(2 * nl, 0.25) if cfg.get("deep_norm", False) else 1.0
        self.eps = 1e-5
        f32 = lambda t: t.float().contiguous().to(dev)
        self.tables = fbank_tables(dev)
        self.patch = make_lin(sd["patch_embedding.weight"].reshape(E, 256), sd.get("patch_embedding.bias"), dtype, dev)
        self.ln0 = (f32(sd["layer_norm.weight"]), f32(sd["layer_norm.bias"]))
        if "post_extract_proj.weight" in sd:
            self.post = make_lin(sd["post_extract_proj.weight"], sd["post_extract_proj.bias"], dtype, dev)
        else:
            assert E == C
            self.post = make_lin(torch.eye(C), None, dtype, dev)
        # conv positional embedding: group g -> W_g [cg, kk*cg] with k = j*cg + ci  (w[g*cg+co, ci, j])
        w = _pos_conv_weight(sd)                                           # [C, cg, kk]
        self.conv_w = [w[g * self.cg:(g + 1) * self.cg].permute(0, 2, 1).reshape(self.cg, self.kk * self.cg)
                       .to(dtype).contiguous().to(dev) for g in range(self.G)]
        self.conv_b = f32(sd["encoder.pos_conv.0.bias"])
        self.enc_ln = (f32(sd["encoder.layer_norm.weight"]), f32(sd["encoder.layer_norm.bias"]))
        self.rel = cfg.get("relative_position_embedding", False)
        self.gru = cfg.get("gru_rel_pos", False)
        if self.rel:
            self.rel_emb = sd["encoder.layers.0.self_attn.relative_attention_bias.weight"].float().cpu()
        inv = 1.0 / self.alpha
        self.layers = []
        for i in range(nl):
            p = "encoder.layers.%d." % i
            a = p + "self_attn."
            Lr = _Layer()
            Lr.qkv = stack_lins([(sd[a + n + ".weight"], sd[a + n + ".bias"], None) for n in ("q_proj", "k_proj", "v_proj")],
                                dtype, dev)
            Lr.out = make_lin(sd[a + "out_proj.weight"] * inv, sd[a + "out_proj.bias"] * inv, dtype, dev)
            Lr.fc1 = make_lin(sd[p + "fc1.weight"], sd[p + "fc1.bias"], dtype, dev)
            Lr.fc2 = make_lin(sd[p + "fc2.weight"] * inv, sd[p + "fc2.bias"] * inv, dtype, dev)
            Lr.ln1 = (f32(sd[p + "self_attn_layer_norm.weight"]), f32(sd[p + "self_attn_layer_norm.bias"]))
            Lr.ln2 = (f32(sd[p + "final_layer_norm.weight"]), f32(sd[p + "final_layer_norm.bias"]))
            if self.rel and self.gru:
                gw, gb = sd[a + "grep_linear.weight"].float(), sd[a + "grep_linear.bias"].float()
                Lr.w2 = f32(torch.stack([gw[:4].sum(0), gw[4:].sum(0)], 0))
                Lr.b2 = f32(torch.stack([gb[:4].sum(), gb[4:].sum()]))
                Lr.grep_a = f32(sd[a + "grep_a"].reshape(-1))
            self.layers.append(Lr)
        self._bias = {}
        self._ones_gate = {}

    # ------------------------------------------------------------------------------------------------------------
    def position_bias(self, Lq):
        """[heads, L, L] fp32 on the device (backbone.py:418-429), cached per sequence length."""
        if Lq not in self._bias:
            ctx = torch.arange(Lq)[:, None]
            mem = torch.arange(Lq)[None, :]
            bucket = relative_position_bucket(mem - ctx, self.cfg["num_buckets"], self.cfg["max_distance"])
            self._bias[Lq] = self.rel_emb[bucket].permute(2, 0, 1).contiguous().to(self.dev)
        return self._bias[Lq]

    @staticmethod
    def forward_padding_mask(n_feat, mask):
        """BEATs.py:102-115 on the host: sample (or frame) mask [B, n] -> one flag per feature = all of its share padded."""
        extra = mask.shape[1] % n_feat
        if extra > 0:
            mask = mask[:, :-extra]
        return mask.reshape(mask.shape[0], n_feat, -1).all(-1)

    def extract_features(self, wav, padding_mask=None, keep=None):
        """wav [B, n] fp16/fp32 (amplitude +-1) -> [B, L, C] 16-bit.  padding_mask [B, n] bool (True = padded sample, the
        `audio_wav_mask` of tdc/cambrian_arch.py:1558) or None: reduced to one flag per fbank frame and then per token
        (BEATs.py:142-153); flagged tokens are zeroed before the positional conv (backbone.py:111-112) and excluded as keys
        from every layer's attention (backbone.py:633-643) - their own output rows are still computed and returned, as in the
        reference."""
        pm = None
        if padding_mask is not None:
            pm = torch.as_tensor(padding_mask).bool().cpu()
            if not bool(pm.any()):
                pm = None
        dt, dev = self.dtype, self.dev
        wav = wav.to(dev)
        if wav.dtype not in (torch.float16, torch.float32):
            wav = wav.float()
        wav = wav.contiguous()
        B = wav.shape[0]
        C, Cp, G, cg, kk = self.C, pad64(self.C), self.G, self.cg, self.kk
        patches, plain, m = ops.fbank(wav, self.tables, dt, want_plain=keep is not None, mean=FBANK_MEAN, std=FBANK_STD)
        Lq = (m // 16) * 8
        R = B * Lq
        tok_mask = rows_masked = None
        if pm is not None:
            assert tuple(pm.shape) == tuple(wav.shape), "padding_mask must match wav"
            tm = self.forward_padding_mask(Lq, self.forward_padding_mask(m, pm))           # [B, Lq] on the host
            if bool(tm.any()):
                assert Lq % 4 == 0, "key padding mask: token count must be a multiple of 4"
                tok_mask = tm.to(torch.uint8).contiguous().to(dev)
                rows_masked = torch.nonzero(tm.reshape(-1)).flatten().to(dev)             # token rows b*Lq + t
        f = ops.gemm(patches, self.patch.w, self.patch.b)                                # [R, pad64(E)]
        f16, _ = ops.layernorm(f, self.ln0[0], self.ln0[1], self.eps, self.E, dt)
        # ---- conv positional embedding
        pad, Lp = kk // 2, Lq + kk
        cmap = (Lq, Lp, pad, 1)                                                          # token (b, t) -> row b*Lp + pad + t
        x32 = ops.gemm(f16, self.post.w, self.post.b, out_f32=True)                      # residual of the conv block
        rows_padded = None
        if rows_masked is not None:                                                      # x[padding_mask] = 0
            x32.index_fill_(0, rows_masked, 0.0)
            rows_padded = (rows_masked // Lq) * Lp + pad + rows_masked % Lq
        xg = torch.zeros(G, B * Lp, cg, device=dev, dtype=dt)                            # zero rows = conv padding
        ypad = torch.empty(B * Lp, Cp, device=dev, dtype=dt)
        Mc = B * Lp - kk + 1
        for g in range(G):
            ops.gemm(f16, self.post.w[g * cg:(g + 1) * cg], self.post.b[g * cg:(g + 1) * cg], out=xg[g], c_map=cmap)
            if rows_padded is not None:
                xg[g].index_fill_(0, rows_padded, 0.0)
            a = torch.as_strided(xg[g], (Mc, kk * cg), (cg, 1))                          # overlapping windows
            ops.gemm(a, self.conv_w[g], self.conv_b[g * cg:(g + 1) * cg], act=L.ACT_GELU_ERF,
                     out=ypad[:, g * cg:(g + 1) * cg], M=Mc)
        # x + gelu(conv(x)) -> LayerNorm: window (b, t) starts at padded row b*Lp + t
        x16, _ = ops.layernorm(ypad, self.enc_ln[0], self.enc_ln[1], self.eps, C, dt, rows=R, x_map=(Lq, Lp, 0, 1),
                               add=x32, add_period=R, add_mode=0)
        # ---- layers
        bias = self.position_bias(Lq) if self.rel else None
        if tok_mask is not None and bias is None:
            raise NotImplementedError("key padding mask without the relative position bias (tdc_attention: biased form only)")
        eps2 = self.eps / (self.alpha * self.alpha)
        ldq = None
        for Lr in self.layers:
            qkv = ops.gemm(x16, Lr.qkv.w, Lr.qkv.b)
            ldq = qkv.stride(0)
            gate = None
            if bias is not None:
                if self.gru:
                    gate = ops.relpos_gate(qkv, R, self.heads, self.hd, Lr.w2, Lr.b2, Lr.grep_a)
                else:
                    gate = self._ones_gate.setdefault(R, torch.ones(R, self.heads, device=dev, dtype=torch.float32))
            attn = torch.empty(R, Cp, device=dev, dtype=dt)
            if Cp != C:
                attn.zero_()
            ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:3 * C], attn, B, self.heads, self.hd, Lq, Lq,
                          self.hd ** -0.5, Lq * ldq, Lq * ldq, Lq * ldq, Lq * Cp, bias=bias, gate=gate,
                          key_mask=tok_mask if bias is not None else None)
            y32 = ops.gemm(attn, Lr.out.w, Lr.out.b, res=x16, out_f32=True)
            x16, _ = ops.layernorm(y32, Lr.ln1[0], Lr.ln1[1], eps2, C, dt)
            h = ops.gemm(x16, Lr.fc1.w, Lr.fc1.b, act=L.ACT_GELU_ERF)
            y32 = ops.gemm(h, Lr.fc2.w, Lr.fc2.b, res=x16, out_f32=True)
            x16, _ = ops.layernorm(y32, Lr.ln2[0], Lr.ln2[1], eps2, C, dt)
        if keep is not None:
            keep.update(fbank=plain, frames=m)
        return x16[:, :C].reshape(B, Lq, C)

    @staticmethod
    def window_starts(n_samples, dist=10):
        return list(range(0, int(n_samples / SAMPLE_RATE), dist))

    def window_token_counts(self, n_samples, dist=10):
        """tokens each 10-second window of an n_samples waveform yields (8 per 16 fbank frames): known without running
        the encoder, so a rank of the frame-sharded path can lay out the audio plan of the whole video."""
        out = []
        for k in self.window_starts(n_samples, dist):
            n = min(n_samples, int(SAMPLE_RATE * (k + dist))) - SAMPLE_RATE * k
            out.append((int(L.load().tdc_fbank_frames(n)) // 16) * 8)
        return out

    def window_features(self, wav, dist=10, only=None, mask=None):
        """the per-window loop of tdc/cambrian_arch.py:1552-1560 for one video: wav [1, N] -> list of [1, L_w, C].
        All full 10-second windows go through one batched call, a shorter last window through a second one.
        only = iterable of window indices: just those windows are encoded and a dict {window: [1, L_w, C]} comes back
        (every op of the encoder is row- / item-wise, so a window's features do not depend on what shares its batch).
        mask [1, N] bool = the `audio_wav_mask` (True = padded sample), sliced per window like the waveform
        (cambrian_arch.py:1558)."""
        assert wav.dim() == 2 and wav.shape[0] == 1
        if mask is not None:
            mask = torch.as_tensor(mask).bool().cpu()
            assert tuple(mask.shape) == tuple(wav.shape)
            if not bool(mask.any()):
                mask = None
        N = wav.shape[1]
        starts = self.window_starts(N, dist)
        sel = list(range(len(starts))) if only is None else sorted(set(int(w) for w in only))
        assert all(0 <= w < len(starts) for w in sel)
        n = SAMPLE_RATE * dist
        full = [w for w in sel if SAMPLE_RATE * (starts[w] + dist) <= N]
        out = {}
        if full:
            if only is None:
                batch = wav[0, : len(full) * n].reshape(len(full), n)       # full windows are contiguous from 0
            else:
                batch = torch.stack([wav[0, SAMPLE_RATE * starts[w]: SAMPLE_RATE * starts[w] + n] for w in full], 0)
            bm = None
            if mask is not None:
                bm = torch.stack([mask[0, SAMPLE_RATE * starts[w]: SAMPLE_RATE * starts[w] + n] for w in full], 0)
            feats = self.extract_features(batch, padding_mask=bm)
            for i, w in enumerate(full):
                out[w] = feats[i:i + 1]
        for w in sel:
            if w not in out:
                k = starts[w]
                a, b = SAMPLE_RATE * k, int(SAMPLE_RATE * (k + dist))
                out[w] = self.extract_features(wav[:, a:b], padding_mask=None if mask is None else mask[:, a:b])
        if only is None:
            return [out[w] for w in sel]
        return out
