"""Pins oracle/tdc_oracle.py to the reference: every stage against fixtures produced by running the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import synth
from util import oracle, load_fixture, embed_fn, pipeline_cfg

ATOL = 2e-5


def close(a, b, atol=ATOL):
    a = torch.as_tensor(a, dtype=torch.float32)
    b = torch.as_tensor(b, dtype=torch.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())  # fp32 round-off grows with magnitude
    assert err <= atol * scale, (err, scale)


def test_bilinear_bicubic_matrices_match_torch():
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    for n_in, n_out in [(27, 24), (9, 8), (6, 8), (37, 27), (5, 9)]:
        x = torch.randn(2, 3, n_in, n_in, generator=g)
        tok = x.permute(0, 2, 3, 1).reshape(2, n_in * n_in, 3)
        ref = F.interpolate(x, size=(n_out, n_out), mode="bilinear", align_corners=False)
        close(oracle.resize_tokens(tok, n_out), ref.permute(0, 2, 3, 1).reshape(2, -1, 3), 3e-5)
        ref = F.interpolate(x, size=(n_out, n_out), mode="bicubic", align_corners=False)
        close(oracle.resize_tokens(tok, n_out, oracle.bicubic_matrix), ref.permute(0, 2, 3, 1).reshape(2, -1, 3), 3e-5)


def test_adaptive_pool_matches_torch():
    import torch.nn.functional as F
    x = torch.randn(156, 7)
    for K in (16, 144, 5):
        ref = F.adaptive_avg_pool1d(x.t()[None], K)[0].t()
        close(oracle.adaptive_avg_pool_tokens(x, K), ref, 1e-6)


def test_siglip_tower():
    W, o = load_fixture("siglip_small.npz")
    out, pre = oracle.siglip_tower(torch.from_numpy(o["pixels"]), W, heads=4, interp_tokens=64)
    close(pre, o["out_pre_interp"])
    close(out, o["out"])


def test_dino_tower():
    W, o = load_fixture("dino_small.npz")
    out, pre = oracle.dino_tower(torch.from_numpy(o["pixels"]), W, heads=4, interp_tokens=64)
    close(pre, o["out_pre_interp"])
    close(out, o["out"])


def test_sva_and_masks():
    W, o = load_fixture("sva_small.npz")
    aux = [torch.from_numpy(o["aux0"]), torch.from_numpy(o["aux1"])]
    sizes = [tuple(int(v) for v in s) for s in o["image_sizes"]]
    out, masks = oracle.sva(aux, torch.from_numpy(o["vision_query"])[0], sizes, W, side=4)
    assert np.array_equal(masks[0].numpy(), o["out_mask0"])
    assert np.array_equal(masks[1].numpy(), o["out_mask1"])
    assert not o["out_mask0"].all()  # the fixture really exercises padding masks
    close(out, o["out"])


def test_qformer_chunk():
    W, o = load_fixture("qformer_small.npz")
    chunk = torch.from_numpy(o["chunk"])
    K = int(o["K"])
    ids = torch.from_numpy(o["prompt_ids"])
    q = oracle._lin(oracle.adaptive_avg_pool_tokens(chunk[0], K), W, "query_proj")
    close(q, o["out_query_tokens"])
    L = chunk.shape[0] - 1
    last = oracle.qformer_bert(q[None].expand(L, -1, -1), chunk[1:], ids, W, heads=4)
    close(last, o["out_last_hidden"])
    close(oracle.qformer_bert(q[None].expand(L, -1, -1), chunk[1:], None, W, heads=4), o["out_last_hidden_notext"])
    comp, _ = oracle.compress_chunk(chunk, ids, W, K, heads=4)
    close(comp, o["out_compressed"])


def _run_pipeline(name):
    W, o = load_fixture(name)
    cfg = pipeline_cfg(o)
    W["embed_tokens_fn"] = embed_fn(o)
    vid = torch.from_numpy(synth.video_from_basis(o["video_basis"], o["video_coef"]))
    r = oracle.encode_video(W, cfg, vid, vid + 0.01, tuple(int(v) for v in o["image_size"]),
                            torch.from_numpy(o["input_ids"]), torch.from_numpy(o["prompt_ids"]))
    return r, o


@pytest.mark.parametrize("name,stride", [("pipeline_T40.npz", 1), ("pipeline_T10_land.npz", 1),
                                         ("pipeline_T260.npz", 16)])
def test_full_pipeline(name, stride):
    r, o = _run_pipeline(name)
    # integers: bit-exact
    assert np.array_equal(r["seg_indices"].numpy(), o["out_seg_indices"])
    assert np.array_equal(r["selected"].numpy(), o["out_selected"])
    assert [list(s) for s in r["final_size"]] == o["out_final_size"].tolist()
    # stages
    close(r["siglip_feat"][::stride], o["out_siglip_feat"])
    close(r["dino_feat"][::stride], o["out_dino_feat"])
    close(r["aux0"][::stride], o["out_aux0"])
    close(r["aux1"][::stride], o["out_aux1"])
    nq = r["sva"].shape[1]
    close(r["sva"][::stride], torch.from_numpy(o["out_sva"]).reshape(-1, nq, r["sva"].shape[2]))
    close(r["mm_proj"][::stride], o["out_mm_proj"])
    close(r["inputs_embeds"], o["out_inputs_embeds"])


@pytest.mark.parametrize("name", ["pipeline_T40_nostatic.npz", "pipeline_T40_learned.npz"])
def test_pipeline_config_ablations(name):
    """add_static=False (every frame compressed, no static tokens) and query_type='learned' (query_tokens), generated
    from the reference with those config values (tdc/cambrian_arch.py:1509-1511,1625-1640,1668-1692)."""
    r, o = _run_pipeline(name)
    assert np.array_equal(r["seg_indices"].numpy(), o["out_seg_indices"])
    close(r["inputs_embeds"], o["out_inputs_embeds"])
    if "nostatic" in name:
        n_text = o["input_ids"].shape[1] - 1
        assert r["inputs_embeds"].shape[1] == n_text + 40 * (4 + 1)


def test_pipeline_token_accounting():
    """T=40 -> static frames emit N+1 tokens, compressed frames K+1."""
    r, o = _run_pipeline("pipeline_T40.npz")
    chunks = oracle.chunk_table(40, r["seg_indices"])
    n_static = len(chunks)
    n_comp = 40 - n_static
    N = 4 * 5  # 4x4 tokens + newline column
    assert r["visual_tokens"].shape[0] == n_static * (N + 1) + n_comp * (4 + 1)
    assert int(o["n_qformer_calls"]) == sum(1 for s, e in chunks if e - s > 1)


def test_full_pipeline_audio():
    """a20: audio tokens interleaved into the Q-Former KV (N+50) and into the static frames (fake BEATs features)."""
    W, o = load_fixture("pipeline_T40_audio.npz")
    cfg = pipeline_cfg(o)
    assert cfg["audio_input"]
    W["embed_tokens_fn"] = embed_fn(o)
    vid = torch.from_numpy(synth.video_from_basis(o["video_basis"], o["video_coef"]))
    wins = synth.beats_windows(torch.from_numpy(o["audio_wav"].astype(np.float32)))
    assert [w.shape[1] for w in wins] == [496] * 4
    r = oracle.encode_video(W, cfg, vid, vid + 0.01, tuple(int(v) for v in o["image_size"]),
                            torch.from_numpy(o["input_ids"]), torch.from_numpy(o["prompt_ids"]), beats_windows=wins)
    assert np.array_equal(r["seg_indices"].numpy(), o["out_seg_indices"])
    close(r["inputs_embeds"], o["out_inputs_embeds"])
    # static frames carry N + 50 + 1 tokens, compressed ones K + 1
    chunks = oracle.chunk_table(40, r["seg_indices"])
    assert r["visual_tokens"].shape[0] == len(chunks) * (20 + 50 + 1) + (40 - len(chunks)) * 5


def test_audio_tokens_with_dropped_frames():
    """seconds dropped by the frame cap are pooled into the preceding kept second (cambrian_arch.py:1570-1589)."""
    g = torch.Generator().manual_seed(0)
    wins = [torch.randn(1, 496, 8, generator=g), torch.randn(1, 300, 8, generator=g)]   # 10 s + 6 s
    samp = torch.tensor([1, 1, 0, 0, 1, 0, 1, 1, 1, 0, 0, 1, 1, 0, 1, 1])
    out = oracle.audio_tokens(wins, samp, int(samp.sum()))
    assert out.shape == (int(samp.sum()), 50, 8)
    close(out[0], wins[0][0, :50])                                              # kept, next kept -> verbatim
    close(out[1], torch.nn.functional.adaptive_avg_pool2d(wins[0][:, 50:200], (50, 8))[0])   # 1 kept + 2 dropped
    close(out[2], torch.nn.functional.adaptive_avg_pool2d(wins[0][:, 200:300], (50, 8))[0])
