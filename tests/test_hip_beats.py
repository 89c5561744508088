"""GPU parity of the BEATs audio encoder against the oracle / the reference-generated fixture."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from test_beats import BO, load_beats_fixture

pytestmark = pytest.mark.gpu


def _enc(W, cfg, dtype=torch.float16):
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.beats import BeatsEncoder
    return BeatsEncoder(W, cfg, dtype=dtype, device="cuda:0")


def rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max()) / max(1e-6, float(b.abs().max()))


def test_fbank_vs_oracle():
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd import ops
    from tdc_video_amd.beats import fbank_tables
    g = torch.Generator().manual_seed(1)
    n = 16000 * 3 + 123
    t = torch.arange(n) / 16000.0
    wav = torch.stack([0.3 * torch.sin(2 * math.pi * (200 + 500 * t) * t) + 0.02 * torch.randn(n, generator=g),
                       0.1 * torch.randn(n, generator=g)]).half()
    tabs = fbank_tables("cuda:0")
    want = BO.preprocess(wav.float())
    patches, plain, m = ops.fbank(wav.cuda(), tabs, torch.float16, want_plain=True)
    assert m == want.shape[1] and tuple(plain.shape) == tuple(want.shape)
    err = (plain.cpu() - want).abs()
    # fp32 FFT / summation-order noise only: normalised log-mel values are O(1)
    assert float(err.max()) < 2e-3 and float(err.mean()) < 2e-5, (float(err.max()), float(err.mean()))
    # fp32 input gives the same as fp16 input holding the same values
    _, plain32, _ = ops.fbank(wav.float().cuda(), tabs, torch.float16, want_plain=True)
    assert torch.equal(plain32, plain)
    # the patch layout is the im2col of the 16x16 / stride 16 conv of the (16-bit cast) plain fbank
    ty = m // 16
    im = plain[:, : ty * 16].reshape(2, ty, 16, 8, 16).permute(0, 1, 3, 2, 4).reshape(2 * ty * 8, 256)
    assert torch.equal(patches, im.half())
    # oracle tables == product tables
    assert torch.equal(tabs[2].cpu()[:, :256], BO.mel_banks())


def test_relpos_gate_and_biased_attention():
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd import ops
    g = torch.Generator().manual_seed(3)
    for (B, H, d, S, dt) in [(2, 12, 64, 496, torch.float16), (3, 4, 16, 48, torch.bfloat16), (1, 12, 64, 360, torch.bfloat16),
                             (2, 4, 32, 200, torch.float16)]:
        C = H * d
        qkv = (torch.randn(B * S, 3 * C, generator=g) * 0.7).to(dt).cuda()
        w2 = torch.randn(2, d, generator=g) * 0.3
        b2 = torch.randn(2, generator=g)
        ga = torch.rand(H, generator=g) + 0.5
        bias = torch.randn(H, S, S, generator=g) * 1.5
        gate = ops.relpos_gate(qkv, B * S, H, d, w2.cuda().contiguous(), b2.cuda(), ga.cuda())
        qh = qkv[:, :C].float().cpu().view(B, S, H, d).transpose(1, 2)
        s = torch.sigmoid(qh @ w2.t() + b2)
        want_gate = (s[..., 0] * (s[..., 1] * ga[None, :, None] - 1.0) + 2.0)               # [B, H, S]
        assert float((gate.cpu().view(B, S, H).permute(0, 2, 1) - want_gate).abs().max()) < 1e-5
        out = torch.zeros(B * S, C, dtype=dt, device="cuda")
        ld = qkv.stride(0)
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, B, H, d, S, S, d ** -0.5, S * ld, S * ld, S * ld,
                      S * C, bias=bias.cuda().contiguous(), gate=gate)
        kh = qkv[:, C:2 * C].float().cpu().view(B, S, H, d).transpose(1, 2)
        vh = qkv[:, 2 * C:].float().cpu().view(B, S, H, d).transpose(1, 2)
        sc = (qh @ kh.transpose(-1, -2)) * d ** -0.5 + want_gate[..., None] * bias[None]
        want = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(B * S, C)
        tol = 2e-3 if dt == torch.float16 else 1.2e-2
        assert rel(out, want) < tol, (B, H, d, S, rel(out, want))
        # key padding mask (backbone.py:633-643): a padded tail per batch item + one isolated key, masked before the bias
        km = torch.zeros(B, S, dtype=torch.bool)
        for bi in range(B):
            km[bi, S - min(5 + 37 * bi, S // 3):] = True
        km[0, 3] = True
        if B > 1:
            km[1, : 70 if S > 200 else 9] = True                  # (S > 200) a whole leading K/V tile masked: running max starts at -inf
        outm = torch.zeros(B * S, C, dtype=dt, device="cuda")
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], outm, B, H, d, S, S, d ** -0.5, S * ld, S * ld, S * ld,
                      S * C, bias=bias.cuda().contiguous(), gate=gate, key_mask=km.cuda())
        scm = sc.masked_fill(km[:, None, None, :], float("-inf"))
        wantm = (torch.softmax(scm, -1) @ vh).transpose(1, 2).reshape(B * S, C)
        assert rel(outm, wantm) < tol, (B, H, d, S, rel(outm, wantm))
        assert not torch.equal(outm, out)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_tiny_encoder_vs_reference_fixture(tag):
    W, cfg, o = load_beats_fixture()
    enc = _enc(W, cfg)
    wav = torch.from_numpy(o["wav_" + tag])                       # fp16, as the reference's collator hands it over
    keep = {}
    out = enc.extract_features(wav, padding_mask=torch.zeros(wav.shape, dtype=torch.bool), keep=keep)
    ref = torch.from_numpy(o["out_" + tag])
    assert tuple(out.shape) == tuple(ref.shape)
    assert float((keep["fbank"].cpu() - torch.from_numpy(o["fbank_" + tag])).abs().max()) < 2e-3
    assert rel(out, ref) < 1e-2, rel(out, ref)


def test_padded_batch_vs_reference_fixture():
    """extract_features with a padding mask (BEATs.py:142-153, backbone.py:111-112,633-643): the reference's output on a
    padded 2-sample batch (tests/golden/make_golden_beats.py)."""
    W, cfg, o = load_beats_fixture()
    enc = _enc(W, cfg)
    wav = torch.from_numpy(o["wav_pad"])
    mask = torch.from_numpy(o["mask_pad"])
    out = enc.extract_features(wav, padding_mask=mask)
    ref = torch.from_numpy(o["out_pad"])
    tokm = torch.from_numpy(o["tokmask_pad"])
    assert tuple(out.shape) == tuple(ref.shape) and bool(tokm.any())
    assert torch.equal(enc.forward_padding_mask(ref.shape[1], enc.forward_padding_mask(
        BO.preprocess(wav.float()).shape[1], mask)), tokm)
    valid = ~tokm
    err = float((out.float().cpu() - ref)[valid].abs().max()) / float(ref[valid].abs().max())
    assert err < 1e-2, err
    # the mask matters: without it the valid tokens of the padded item come out different
    plain = enc.extract_features(wav)
    assert float((plain.float().cpu() - ref)[valid].abs().max()) / float(ref[valid].abs().max()) > 2 * err
    # the unpadded item of the batch is the unmasked computation
    b_full = int(torch.nonzero(~tokm.any(1))[0])
    assert torch.equal(out[b_full], plain[b_full])


def test_beats_handle_returns_token_mask():
    """model.BeatsHandle.extract_features mirrors BEATs.extract_features(feature_only=True): (features, token-level mask)."""
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.model import BeatsHandle
    W, cfg, o = load_beats_fixture()
    h = BeatsHandle(cfg, W)
    wav, mask = torch.from_numpy(o["wav_pad"]), torch.from_numpy(o["mask_pad"])
    feats, pm = h.extract_features(wav, padding_mask=mask, feature_only=True)
    assert torch.equal(pm, torch.from_numpy(o["tokmask_pad"]))
    assert torch.equal(feats, _enc(W, cfg).extract_features(wav, padding_mask=mask))
    feats0, pm0 = h.extract_features(wav)
    assert pm0 is None and tuple(feats0.shape) == tuple(feats.shape)
    with pytest.raises(NotImplementedError):
        h.extract_features(wav, feature_only=False)


def test_window_features_with_padding_mask():
    """the per-window loop of cambrian_arch.py:1552-1560 slices audio_wav_mask like the waveform: a padded tail in the last
    full window and in the short window; each window equals the direct call on its slice."""
    W, cfg, o = load_beats_fixture()
    enc = _enc(W, cfg)
    g = torch.Generator().manual_seed(5)
    N = 16000 * 23 + 800
    wav = (0.1 * torch.randn(1, N, generator=g)).half()
    mask = torch.zeros(1, N, dtype=torch.bool)
    mask[0, 16000 * 17: 16000 * 20] = True
    mask[0, 16000 * 22:] = True
    got = enc.window_features(wav, mask=mask)
    assert len(got) == 3
    for w, (a, b) in enumerate([(0, 160000), (160000, 320000), (320000, N)]):
        want = enc.extract_features(wav[:, a:b], padding_mask=mask[:, a:b])
        assert torch.equal(got[w], want)
    assert torch.equal(got[0], enc.window_features(wav)[0])
    sel = enc.window_features(wav, only=[1], mask=mask)
    assert torch.equal(sel[1], got[1])


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-2), (torch.bfloat16, 6e-2)])
def test_full_size_encoder_vs_oracle(dtype, tol):
    """BEATs iter3 dimensions (12 x 768, 496 tokens per 10-s window), random weights at natural scale."""
    cfg = dict(BO.BEATS_ITER3_CFG)
    g = torch.Generator().manual_seed(21)
    C, E, F, Hh, nl = 768, 512, 3072, 12, cfg["encoder_layers"]
    rn = lambda *s, std=0.02: torch.randn(*s, generator=g) * std
    W = {"patch_embedding.weight": rn(E, 1, 16, 16, std=0.06), "layer_norm.weight": 1 + rn(E, std=0.1),
         "layer_norm.bias": rn(E, std=0.1), "post_extract_proj.weight": rn(C, E, std=0.04),
         "post_extract_proj.bias": rn(C), "encoder.pos_conv.0.weight_g": 1 + rn(1, 1, 128, std=0.2).abs(),
         "encoder.pos_conv.0.weight_v": rn(C, C // 16, 128, std=0.02), "encoder.pos_conv.0.bias": rn(C),
         "encoder.layer_norm.weight": 1 + rn(C, std=0.1), "encoder.layer_norm.bias": rn(C, std=0.1)}
    emb = rn(320, Hh, std=1.0)
    for i in range(nl):
        p = "encoder.layers.%d." % i
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            W[p + "self_attn." + n + ".weight"] = rn(C, C, std=0.04)
            W[p + "self_attn." + n + ".bias"] = rn(C)
        W[p + "self_attn.grep_linear.weight"] = rn(8, 64, std=0.2)
        W[p + "self_attn.grep_linear.bias"] = rn(8, std=0.2)
        W[p + "self_attn.grep_a"] = (1 + rn(1, Hh, 1, 1, std=0.2))
        W[p + "self_attn.relative_attention_bias.weight"] = emb
        W[p + "fc1.weight"], W[p + "fc1.bias"] = rn(F, C, std=0.04), rn(F)
        W[p + "fc2.weight"], W[p + "fc2.bias"] = rn(C, F, std=0.03), rn(C)
        for n in ("self_attn_layer_norm", "final_layer_norm"):
            W[p + n + ".weight"], W[p + n + ".bias"] = 1 + rn(C, std=0.1), rn(C, std=0.1)
    n = 160000
    t = torch.arange(n) / 16000.0
    wav = torch.stack([0.2 * torch.sin(2 * math.pi * (100 + 40 * t) * t) + 0.05 * torch.randn(n, generator=g),
                       0.1 * torch.randn(n, generator=g)]).half()
    enc = _enc(W, cfg, dtype)
    out = enc.extract_features(wav)
    assert tuple(out.shape) == (2, 496, 768)
    want = BO.extract_features(W, cfg, wav.float())
    assert rel(out, want) < tol, rel(out, want)
    # window loop: 2 full windows batched + a 3.5-s tail, same values as window-by-window calls
    long = torch.cat([wav[0], wav[1], wav[0][:56000]])[None]
    wins = enc.window_features(long)
    assert [tuple(w.shape) for w in wins] == [(1, 496, 768), (1, 496, 768), (1, (tdc_frames(56000) // 16) * 8, 768)]
    assert torch.equal(wins[0][0], out[0]) and torch.equal(wins[1][0], out[1])
    tail = BO.extract_features(W, cfg, long[:, 320000:].float())
    assert rel(wins[2], tail) < tol


def tdc_frames(n):
    return 1 + (n - 400) // 160
