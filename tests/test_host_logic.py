"""CPU: the product's host integer logic (tdc-video_amd/segment.py) against the oracle (itself pinned to the reference),
and the drop-in boundary (class / method / state-dict names of tdc-video_amd/model.py) against the golden fixtures."""
import inspect
import random
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from util import oracle, load_fixture

import tdc_video_amd  # noqa: F401
from tdc_video_amd import segment as seg


def test_frame_budget_and_uniform_indices():
    for T in (1, 8, 26, 224, 225, 260, 512, 1000):
        for mlen, K in ((8192, 16), (2100, 4), (8192, 144), (4096, 16)):
            cfg = dict(tokenizer_model_max_length=mlen, context_token_num=K)
            ids = torch.arange(9)
            mx = oracle.get_max_num_frames(ids, cfg)
            assert seg.get_max_num_frames(9, cfg) == mx
            idx, samp = oracle.frame_cap_indices(T, mx)
            assert seg.uniform_indices(T, min(mx, 224)) == idx
            assert int(samp.sum()) == len(idx)
    cfg = dict(tokenizer_model_max_length=8192, context_token_num=16, audio_input=True)
    assert seg.get_max_num_frames(100, cfg) == (8192 - 100 - 16) // ((144 + 50 + 112) // 8)
    ids = torch.tensor([5, 6, 151643, 7])
    assert oracle.get_max_num_frames(ids, dict(tokenizer_model_max_length=8192)) == seg.get_max_num_frames(2, dict(
        tokenizer_model_max_length=8192))


def test_segment_selection_and_chunks():
    rng = random.Random(0)
    for T in (26, 40, 97, 224, 511):
        sims = [rng.random() for _ in range(T - 1)]
        want = oracle.select_segments(torch.tensor(sims), 24).tolist()
        got = seg.select_segments(sims, 24)
        assert got == want
        assert seg.chunk_table(T, got) == oracle.chunk_table(T, torch.tensor(want))
    # ties resolve to the lowest index (stable)
    assert seg.select_segments([0.5] * 30, 24) == list(range(24))
    # T <= 25: every frame its own segment
    assert seg.chunk_table(10, list(range(10))) == [(i, i + 1) for i in range(10)]
    assert all(e - s <= 8 for s, e in seg.chunk_table(300, [100]))


@pytest.mark.parametrize("size", [(384, 384), (360, 640), (640, 360), (200, 640), (640, 200), (1080, 1920), (100, 101)])
def test_unpad_geometry_and_masks(size):
    for side, r in ((12, 2), (4, 2), (6, 4)):
        assert seg.unpad_bounds(side, side, size) == oracle.unpad_bounds(side, side, size)
        m = torch.tensor(seg.window_mask_bytes(side, r, size), dtype=torch.bool)
        assert torch.equal(m, oracle.window_masks(side, r, size))
        src, (h, w) = seg.unpad_newline_map(side, size, frame=3)
        assert len(src) == h * (w + 1)
        r0, r1, c0, c1 = oracle.unpad_bounds(side, side, size)
        assert (h, w) == (r1 - r0, c1 - c0)
        assert src[w] == (1, 0) and src[0] == (0, 3 * side * side + r0 * side + c0)


def test_emit_plan_without_static_frames():
    """add_static=False: every frame (key frames and single-frame chunks included) is compressed, K+1 tokens each."""
    T, N, K = 21, 7, 3
    segi = [2, 3, 9]                                         # chunks: [0,3) [3,4) [4,10) [10,18) [18,21)
    plan = seg.emit_plan(T, N, K, segi, 10 ** 9, add_static=False)
    assert plan["chunks"] == [(0, 3), (3, 4), (4, 10), (10, 18), (18, 21)]
    assert plan["comp_frames"] == list(range(T))
    assert plan["key_frames"] == [0, 3, 4, 10, 18]
    assert plan["comp_chunk"] == [0] * 3 + [1] + [2] * 6 + [3] * 8 + [4] * 3
    assert all(e[0] != "f" for e in plan["src"]) and len(plan["src"]) == T * (K + 1)
    assert plan["src"][:K + 1] == [("c", 0, 0), ("c", 0, 1), ("c", 0, 2), ("s",)]
    # tail clipping removes ceil(excess / n_chunks) tokens from the end of every chunk's block
    clipped = seg.emit_plan(T, N, K, segi, T * (K + 1) - 7, add_static=False)
    assert len(clipped["src"]) == T * (K + 1) - 2 * 5
    # default unchanged
    assert seg.emit_plan(T, N, K, segi, 10 ** 9) == seg.emit_plan(T, N, K, segi, 10 ** 9, add_static=True)


def test_emit_plan_layout_and_clipping():
    T, N, K = 40, 20, 4
    segi = [1, 2, 4, 5, 7, 8, 11, 13, 14, 16, 17, 18, 19, 20, 22, 23, 24, 25, 26, 29, 32, 34, 35, 38]
    plan = seg.emit_plan(T, N, K, segi, 10 ** 9)
    chunks = plan["chunks"]
    n_static = len(chunks)
    assert len(plan["src"]) == n_static * (N + 1) + (T - n_static) * (K + 1)
    assert plan["comp_frames"] == [f for s, e in chunks for f in range(s + 1, e)]
    assert plan["key_frames"] == [s for s, e in chunks if e - s > 1]
    # clipping reproduces tdc/cambrian_arch.py:1694-1709 on a token-id model of the stream
    # ... incl. a zero and NEGATIVE budgets (text longer than the model length): the reference's python slices then drop the
    # last |budget| tokens of what the per-chunk clipping left (cambrian_arch.py:1709)
    for budget in (500, 333, 100, 7, 0, -3, -40):
        ids_per_chunk = []
        for (s, e) in chunks:
            t = [("f", s, i) for i in range(N)] + [("s",)]
            for f in range(s + 1, e):
                t += [("c", plan["comp_frames"].index(f), k) for k in range(K)] + [("s",)]
            ids_per_chunk.append(t)
        total = sum(len(t) for t in ids_per_chunk)
        if total > budget:
            import math
            rm = math.ceil((total - budget) / len(ids_per_chunk))
            ids_per_chunk = [t[:-rm] for t in ids_per_chunk]
        want = [x for t in ids_per_chunk for x in t][:budget]
        assert seg.emit_plan(T, N, K, segi, budget)["src"] == want


def test_audio_plan_matches_oracle():
    g = torch.Generator().manual_seed(0)
    wins = [torch.randn(1, 496, 8, generator=g), torch.randn(1, 496, 8, generator=g), torch.randn(1, 300, 8, generator=g)]
    for samp in ([1] * 26, [1, 1, 0, 0, 1, 0, 1, 1, 1, 0] * 2 + [0, 1, 1, 0, 1, 1], [1] + [0] * 24 + [1]):
        want = oracle.audio_tokens(wins, torch.tensor(samp), sum(samp))
        plan = seg.audio_plan([496, 496, 300], samp)
        pool = lambda x: x if x.shape[0] == 50 else torch.nn.functional.adaptive_avg_pool2d(x[None], (50, 8))[0]
        got = []
        for parts, direct in plan:
            toks = [pool(wins[w][0, s:e]) for (w, s, e) in parts]
            got.append(toks[0] if len(toks) == 1 else pool(torch.cat(toks, 0)))
        assert len(got) == sum(samp)
        assert torch.allclose(torch.stack(got), want, atol=1e-6)


def test_shard_ranges():
    for T in (512, 100, 7):
        for w in (1, 2, 4, 8):
            r = seg.shard_ranges(T, w)
            assert r[0][0] == 0 and r[-1][1] == T and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


# ------------------------------------------------------------------------------------------------ boundary mirror
TINY = dict(tdc_tower_archs=dict(siglip=dict(D=48, layers=2, mlp=80, n_pos=81, heads=4),
                                 dino=dict(D=64, layers=2, mlp=176, n_pos=25, heads=4)),
            tdc_qformer_arch=dict(hidden=64, layers=4, heads=4, ffn=128, vocab=300, max_pos=64))


def tiny_config(**over):
    cfg = types.SimpleNamespace(
        mm_vision_tower_aux_list=["siglip/CLIP-ViT-SO400M-14-384", "facebook/dinov2-giant-res378"],
        mm_vision_tower_aux_token_len_list=[64, 64], mm_projector_type="sva", vision_hidden_size=64,
        num_query_group=1, query_num_list=[16], image_token_len=16, connector_only=True, connector_depth=2,
        hidden_size=96, model_type="qwen2", tokenizer_model_max_length=8192, inference_max_length=16,
        tokenizer_padding_side="right", context_token_num=4, **TINY)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def build_stub_lm(cfg):
    from tdc_video_amd.model import CambrianMetaModel, CambrianMetaForCausalLM

    class StubBase(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.config = config
            self.embed_tokens = nn.Embedding(2000, config.hidden_size)

        @property
        def dtype(self):
            return torch.float32

    class StubModel(CambrianMetaModel, StubBase):
        pass

    class StubLM(nn.Module, CambrianMetaForCausalLM):
        def __init__(self, config):
            nn.Module.__init__(self)
            self.config = config
            self.model = StubModel(config)

        def get_model(self):
            return self.model
    return StubLM(cfg)


def test_state_dict_names_match_reference():
    W, o = load_fixture("pipeline_T40.npz")
    lm = build_stub_lm(tiny_config())
    mine = lm.model.tdc_state_dict()
    ref_keys = {k for k in W}
    my_keys = set(mine)
    assert ref_keys <= my_keys, sorted(ref_keys - my_keys)[:5]
    assert my_keys - ref_keys <= set(), sorted(my_keys - ref_keys)[:5]
    for k in ref_keys:
        assert tuple(mine[k].shape) == tuple(W[k].shape), k
    # the nn.Module view uses the reference's 'model.' prefix
    sd = lm.state_dict()
    assert "model.Qformer.bert.encoder.layer.0.crossattention.self.key.weight" in sd
    assert "model.vision_sampler_0.layers.1.cross_attn.k_proj_1.0.bias" in sd
    assert "model.mm_projector_aux_1.3.weight" in sd and "model.frame_seg" in sd


def test_method_signatures_match_reference():
    from tdc_video_amd.model import CambrianMetaForCausalLM, CambrianMetaModel
    sig = inspect.signature(CambrianMetaForCausalLM.prepare_inputs_labels_for_multimodal)
    assert list(sig.parameters) == ["self", "input_ids", "position_ids", "attention_mask", "past_key_values", "labels",
                                    "images", "image_aux_attention_masks_list", "image_sizes", "video_indices",
                                    "prompts", "audios"]  # tdc/cambrian_arch.py:864-877
    assert list(inspect.signature(CambrianMetaForCausalLM.encode_images).parameters) == ["self", "image_aux_list",
                                                                                         "encode_type"]
    assert list(inspect.signature(CambrianMetaForCausalLM.adapt_segment).parameters)[:4] == [
        "self", "feature_list", "split_sizes", "new_image_aux_list"]
    for name in ("get_vision_tower_aux_list", "initialize_vision_modules", "initialize_compressor",
                 "initialize_audio", "get_frame_pos"):
        assert hasattr(CambrianMetaModel, name)
    for name in ("get_model", "get_max_num_frames", "initialize_vision_tokenizer"):
        assert hasattr(CambrianMetaForCausalLM, name)


def test_early_out_returns_inputs_untouched():
    lm = build_stub_lm(tiny_config())
    ids = torch.tensor([[5]])
    out = lm.prepare_inputs_labels_for_multimodal(ids, None, None, "pkv", None, images=[torch.zeros(1)])
    assert out[0] is ids and out[3] == "pkv" and out[4] is None and len(out) == 10
    out = lm.prepare_inputs_labels_for_multimodal(torch.tensor([[5, 6]]), None, None, None, None, images=None)
    assert out[4] is None and len(out) == 10


def test_checkpoint_loader_roundtrip(tmp_path):
    """8(f)-4: a released-checkpoint-style directory (sharded safetensors + index, 'model.' prefixed reference names,
    unrelated LLM tensors mixed in) fills the boundary model; towers load from HF-style directories."""
    from tdc_video_amd import checkpoint as ck
    from util import write_released_style_checkpoint
    W, o = load_fixture("pipeline_T40.npz")
    d, sig_dir, dino_dir = write_released_style_checkpoint(W, tmp_path)
    lm = build_stub_lm(tiny_config())
    missing, unexpected = ck.load_path_weights(lm.model, d, sig_dir, dino_dir)
    assert missing == [] and unexpected == []
    got = lm.model.tdc_state_dict()
    for k, v in W.items():
        assert torch.equal(got[k], v), k


def test_weight_prep_layernorm_fold_and_fp8(monkeypatch):
    """One-time weight preparation of the optional tower paths, on the CPU: the LayerNorm fold (W diag(gamma), bias
    beta W^T + b, c1 = row sums of the ROUNDED folded weight) reproduces LN(x) W^T + b, and the fp8 preparation keeps
    every weight within e4m3's half-ulp of its 16-bit value with the row maximum on the largest e4m3 number."""
    from tdc_video_amd import weights as Wt
    g = torch.Generator().manual_seed(0)
    D, N = 128, 192
    W, b = torch.randn(N, D, generator=g) / D ** 0.5, torch.randn(N, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    x = torch.randn(7, D, generator=g) * 2 + 0.3
    lin = Wt.make_lin(W, b, torch.float32, "cpu", col_scale=gamma, col_shift=beta)
    c1 = Wt.fold_c1(lin)
    mean, rstd = x.mean(1, keepdim=True), (x.var(1, unbiased=False, keepdim=True) + 1e-6).rsqrt()
    got = rstd * (x @ lin.w[:N, :D].t() - mean * c1[:N]) + lin.b[:N]
    ref = torch.nn.functional.layer_norm(x, (D,), gamma, beta, 1e-6) @ W.t() + b
    assert (got - ref).abs().max().item() < 1e-4
    # fp8: per-tensor scale, bytes of e4m3, K padded to whole 128-byte tiles
    lin16 = Wt.make_lin(W, b, torch.bfloat16, "cpu")
    l8 = Wt.to_fp8(lin16, "cpu")
    assert l8.w.dtype == torch.uint8 and l8.w.shape == (192, 128) and l8.zeros.numel() == 192 and l8.b is lin16.b
    deq = l8.w.view(torch.float8_e4m3fn).float() * l8.wscale
    assert (deq - lin16.w.float()).abs().max().item() <= lin16.w.float().abs().max().item() * 2 ** -4
    assert abs(deq.abs().max().item() - lin16.w.float().abs().max().item()) < 1e-6
    # switches: fusion needs whole 64-column slots, fp8 whole 128-byte K tiles, and fp8 excludes the fusion
    assert Wt.ln_fusion_enabled(1152, True) and not Wt.ln_fusion_enabled(48, True)
    assert Wt.fp8_enabled(1536, True) and not Wt.fp8_enabled(1152 + 64, True) and not Wt.fp8_enabled(1536, False)
    assert not Wt.ln_fusion_enabled(1152)


def test_non_reference_config_keys_are_validated():
    """config.tdc_* keys: a typo fails with an error that names the key - not a bare KeyError from a table."""
    from tdc_video_amd import model as M
    assert M._dtype_key({"tdc_tower_dtype": "bf16"}, "tdc_tower_dtype", allow32=False) == torch.bfloat16
    assert M._dtype_key({"tdc_tower_res_dtype": "half"}, "tdc_tower_res_dtype") == torch.float16
    assert M._dtype_key({"tdc_tower_res_dtype": "float32"}, "tdc_tower_res_dtype") is None
    assert M._dtype_key({}, "tdc_dino_dtype", allow32=False) is None
    for key, bad, allow32 in (("tdc_tower_dtype", "float8", False), ("tdc_tower_dtype", "float32", False),
                              ("tdc_tower_res_dtype", "fp64", True), ("tdc_dino_dtype", 16, False)):
        with pytest.raises(ValueError, match=key):
            M._dtype_key({key: bad}, key, allow32=allow32)
    lm = build_stub_lm(tiny_config())
    assert lm.get_model().tdc_frame_cap() == 224                    # the reference's constant (cambrian_arch.py:907-916,813-822)
    lm.get_model().config.tdc_frame_cap = 512
    assert lm.get_model().tdc_frame_cap() == 512
    for bad in (0, -1, "224", 3.5, True):
        lm.get_model().config.tdc_frame_cap = bad
        with pytest.raises(ValueError, match="tdc_frame_cap"):
            lm.get_model().tdc_frame_cap()
    assert lm.get_model().tdc_sharded_engine() is None               # no key, no process group
