"""GPU (-m gpu): BASELINE's headline workload (T = 512, K = 144, H = 3584, full-depth towers) THROUGH the drop-in boundary.

SURVEY D3: the reference hard-codes 224 as the cap of both frame sub-samplings (tdc/cambrian_arch.py:907-916 before the
towers, :813-822 inside adapt_segment).  `config.tdc_frame_cap` makes it a parameter (default 224 = parity), so a caller of
`prepare_inputs_labels_for_multimodal` - not only a caller of VideoEncoder - reaches T = 512.  What is checked here:
  * the mixin with the cap lifted emits the same visual rows as VideoEncoder.encode_video(frame_cap=512), bit for bit, and the
    text rows around them are embed_tokens rows;
  * with the default cap the same call keeps 224 frames and equals encode_video(frame_cap=224);
  * the engine the mixin builds by itself chooses one 512-frame tower batch (VideoEncoder.auto_tower_batch), shrinks it when HBM
    is short, and the batch does not change a bit of the result;
  * the config keys are validated with errors that name them.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PROMPT = [101] + list(range(2000, 2010)) + [102]
T, K, H = 512, 144, 3584


@pytest.fixture(scope="module")
def world():
    import bench
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    cfg = bench.model_cfg(H, K, T)
    sd = bench.random_state_dict(H, K, dev, gen)
    extra = dict(tdc_tower_dtype="bfloat16", tdc_tower_res_dtype="float16", tdc_frame_cap=T)
    lm = bench.build_mixin_lm(cfg, sd, dev, torch.float16, extra)
    enc = VideoEncoder(sd, cfg, dtype=torch.float16, device=dev, tower_dtype=torch.bfloat16, tower_res_dtype=torch.float16,
                       tower_batch=512)
    del sd
    torch.cuda.empty_cache()
    vs = bench.synth_video(0, T, 384, dev, torch.bfloat16)
    vd = bench.synth_video(0, T, 378, dev, torch.bfloat16, seed=4321)
    ids = torch.arange(100, 165, device=dev)
    ids[14] = -200
    return lm, enc, vs, vd, ids[None]


def call(lm, vs, vd, ids):
    with torch.inference_mode():
        return lm.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [vs[None], vd[None]],
                                                       image_sizes=[(384, 384)], video_indices=[None], prompts=[PROMPT],
                                                       audios=[None])


def test_mixin_T512_cap_lifted_equals_engine_bit_for_bit(world):
    lm, enc, vs, vd, ids = world
    eng = lm.get_model().tdc_engine()
    assert eng.tower_batch is None and eng.auto_tower_batch(eng.towers["dino"], vd) == 512
    assert eng.tower_res_dtype == torch.float16 and eng.tower_dtype == torch.bfloat16 and eng.dtype == torch.float16
    out = call(lm, vs, vd, ids)
    emb = out[4]
    info = {}
    want = enc.encode_video(vs, vd, (384, 384), budget_text_len=65, n_text_tokens=64, prompt_ids=PROMPT, frame_cap=T, info=info)
    assert len(info["frame_indices"]) == T and len(info["selected"]) == T            # nothing was sub-sampled
    n = want.shape[0]
    assert emb.shape == (1, 64 + n, H) and len(out[8]) == T
    assert torch.equal(emb[0, 14:14 + n], want)
    table = lm.get_model().embed_tokens.weight
    assert torch.equal(emb[0, :14], table[ids[0, :14]]) and torch.equal(emb[0, 14 + n:], table[ids[0, 15:]])
    # a smaller tower batch (what a short-of-HBM device would choose) does not change a bit
    eng.tower_batch = 200
    assert torch.equal(call(lm, vs, vd, ids)[4], emb)
    eng.tower_batch = None
    # the automatic choice halves the batch until the workspace fits 60 % of what is free
    import ctypes as C
    from tdc_video_amd import lib as L
    d = eng.towers["dino"]
    m = eng._vit_struct(d, 27, 27)[0]

    def ws(B):
        return L.load().tdc_vit_workspace_bytes(C.byref(m), B, 378, 378)
    assert 10e9 < ws(512) < 25e9                                     # ~15 GB for one 512-frame batch
    free = 0.3 * ws(512) / 0.6                                       # 60 % of it holds 0.3 of the full workspace
    b = eng.auto_tower_batch(d, vd, free_bytes=free)
    assert b in (64, 128) and ws(b) <= 0.6 * free < ws(2 * b)
    assert eng.auto_tower_batch(d, vd, free_bytes=2 ** 20) == 1
    assert eng.auto_tower_batch(d, vd[:40]) == 40


def test_mixin_default_cap_is_the_reference_constant(world):
    lm, enc, vs, vd, ids = world
    cfg = lm.get_model().config
    del cfg.tdc_frame_cap
    try:
        out = call(lm, vs, vd, ids)
        info = {}
        want = enc.encode_video(vs, vd, (384, 384), budget_text_len=65, n_text_tokens=64, prompt_ids=PROMPT, info=info)
        assert len(info["frame_indices"]) == 224 and info["frame_indices"][1] == int(512 / 224.0)
        assert len(out[8]) == 224 and torch.equal(out[4][0, 14:14 + want.shape[0]], want)
        # adapt_segment's own cap (:813-822) follows the same key: 300 DINO-feature frames -> 224 kept, or all of them
        feats = torch.randn(300, 576, 64, device=vs.device, dtype=torch.float16)
        px = torch.zeros(300, 1)
        f, sizes, _, sel, segi = lm.adapt_segment(feats, [300], [px, px])
        assert sizes == [224] and sel[0].tolist() == [int(300 / 224.0 * i) for i in range(224)] and len(segi[0]) == 24
        cfg.tdc_frame_cap = 512
        f, sizes, _, sel, segi = lm.adapt_segment(feats, [300], [px, px])
        assert sizes == [300] and sel[0].tolist() == list(range(300))
    finally:
        cfg.tdc_frame_cap = T


def test_config_keys_are_validated(world):
    lm = world[0]
    m = lm.get_model()
    cfg = m.config
    for bad in (0, -5, "512", 2.5, True):
        cfg.tdc_frame_cap = bad
        with pytest.raises(ValueError, match="tdc_frame_cap"):
            m.tdc_frame_cap()
    cfg.tdc_frame_cap = T
    keep = m._tdc_encoder
    try:
        for key, bad in (("tdc_tower_dtype", "float8"), ("tdc_tower_dtype", "float32"), ("tdc_dino_dtype", "double"),
                         ("tdc_tower_res_dtype", "fp64"), ("tdc_tower_batch", -1), ("tdc_tower_batch", "64"),
                         ("tdc_selection_refine", "yes"), ("tdc_selection_eps", 2), ("tdc_selection_eps", "1e-3")):
            old = getattr(cfg, key, None)
            setattr(cfg, key, bad)
            with pytest.raises(ValueError, match=key):
                m.tdc_engine(refresh=True)
            if old is None:
                delattr(cfg, key)
            else:
                setattr(cfg, key, old)
    finally:
        m._tdc_encoder = keep
