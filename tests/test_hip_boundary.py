"""GPU (-m gpu): the drop-in boundary's public methods are CALLED, not just inspected - the mixin's `encode_images` (all
three encode_type values, tdc/cambrian_arch.py:698-745), `adapt_segment` (the 5-tuple, :783-861), string prompts through
`bert_tokenizer` (:1527-1534), a batch of two videos, and the reference's behaviour for several <image> tokens in one
sample - against the reference-generated fixtures."""
import pytest
import torch

import synth
from test_host_logic import build_stub_lm, tiny_config
from util import load_fixture, pipeline_cfg

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.float().cpu(), torch.as_tensor(b).float()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


def loaded_lm(W, o, **cfg_over):
    lm = build_stub_lm(tiny_config(**cfg_over))
    m = lm.model
    missing, unexpected = m.load_state_dict({k: v for k, v in W.items() if not k.startswith("vision_tower_aux_list")},
                                            strict=False)
    assert not unexpected, unexpected
    for i, t in enumerate(m.vision_tower_aux_list):
        pre = "vision_tower_aux_list.%d.vision_tower." % i
        t.load_model(state_dict={k[len(pre):]: v for k, v in W.items() if k.startswith(pre)})
    with torch.no_grad():
        m.embed_tokens.weight[[int(i) for i in o["used_embed_ids"]]] = torch.from_numpy(o["used_embed_rows"])
    return lm


def test_encode_images_all_encode_types_vs_golden():
    W, o = load_fixture("pipeline_T40.npz")
    lm = loaded_lm(W, o)
    vid = torch.from_numpy(synth.video_from_basis(o["video_basis"], o["video_coef"]))
    aux = [vid, vid + 0.01]
    sel = o["out_selected"].tolist()
    d = lm.encode_images(aux, encode_type="dino")
    s = lm.encode_images(aux, encode_type="siglip")
    both = lm.encode_images(aux)
    assert isinstance(both, list) and len(both) == 2
    assert d.shape == (40, 64, 64) and s.shape == (40, 64, 48)
    assert torch.equal(both[0], s) and torch.equal(both[1], d)
    assert rel(d[sel], o["out_dino_feat"]) < 4e-3
    assert rel(s[sel], o["out_siglip_feat"]) < 2.5e-2              # the fixtures' x3-scaled SigLIP tower (test_hip_pipeline)
    # 64-frame batching (cambrian_arch.py:702-745): the result does not depend on the batch size, bit for bit
    eng = lm.get_model().tdc_engine()
    eng.tower_batch = 7
    assert torch.equal(lm.encode_images(aux, encode_type="dino"), d)


@pytest.mark.parametrize("name", ["pipeline_T40.npz", "pipeline_T260.npz", "pipeline_T10_land.npz"])
def test_adapt_segment_five_tuple_vs_golden(name):
    """adapt_segment on the DINO features of the frames that survive a1: (features, split_sizes, [pixels, pixels],
    selected_frame_indices_all, segment_frame_indices_all) as the reference returns them (:851-861)."""
    from tdc_video_amd import segment as seg
    W, o = load_fixture(name)
    lm = loaded_lm(W, o, tokenizer_model_max_length=pipeline_cfg(o)["tokenizer_model_max_length"])
    vid = torch.from_numpy(synth.video_from_basis(o["video_basis"], o["video_coef"]))
    ids = torch.from_numpy(o["input_ids"])
    # a1 first (prepare_inputs_labels_for_multimodal does it before encode_images, :899-935)
    idx = seg.uniform_indices(vid.shape[0], min(lm.get_max_num_frames(ids[0]), 224))
    aux = [vid[idx], (vid + 0.01)[idx]]
    feats = lm.encode_images(aux, encode_type="dino")
    f, sizes, new_aux, sel_all, seg_all = lm.adapt_segment(feats, [len(idx)], aux, max_num_segments=24)
    assert sizes == [len(o["out_selected"])]
    assert sel_all[0].tolist() == o["out_selected"].tolist()
    assert seg_all[0].tolist() == o["out_seg_indices"].tolist()
    assert f.shape[0] == sizes[0] and new_aux[0].shape[0] == sizes[0] and new_aux[1].shape[0] == sizes[0]
    assert torch.equal(new_aux[0], aux[0][sel_all[0]]) and torch.equal(f, feats[sel_all[0]])
    # two videos in one call: split_sizes routes them independently
    f2, sizes2, _, sel2, seg2 = lm.adapt_segment(torch.cat([feats, feats[:9]]), [len(idx), 9],
                                                 [torch.cat([aux[0], aux[0][:9]]), torch.cat([aux[1], aux[1][:9]])])
    assert sizes2 == [sizes[0], 9] and seg2[0].tolist() == seg_all[0].tolist() and seg2[1].tolist() == list(range(9))


class FakeBertTokenizer:
    """stands where BertTokenizer.from_pretrained('./checkpoints/bert-base-uncased') does (cambrian_arch.py:405): maps a
    known prompt string to the fixture's BERT ids, same call signature / return shape"""

    def __init__(self, table):
        self.table = table
        self.calls = []

    def __call__(self, text, padding=None, truncation=None, max_length=None, return_tensors=None):
        self.calls.append(dict(text=text, padding=padding, truncation=truncation, max_length=max_length,
                               return_tensors=return_tensors))

        class Enc:
            pass
        e = Enc()
        e.input_ids = torch.tensor([self.table[text][:max_length]])
        e.to = lambda *_a, **_k: e
        return e


def test_string_prompt_goes_through_bert_tokenizer():
    W, o = load_fixture("pipeline_T40.npz")
    lm = loaded_lm(W, o)
    pid = [int(i) for i in o["prompt_ids"]]
    tok = FakeBertTokenizer({"what happens in the video?": pid})
    lm.get_model().bert_tokenizer = tok
    vid = torch.from_numpy(synth.video_from_basis(o["video_basis"], o["video_coef"]))
    images = [vid.unsqueeze(0), (vid + 0.01).unsqueeze(0)]
    ids = torch.from_numpy(o["input_ids"])
    kw = dict(image_sizes=[tuple(int(v) for v in o["image_size"])], video_indices=[None], audios=[None])
    a = lm.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, images,
                                                prompts=["what happens in the video?"], **kw)
    b = lm.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, images, prompts=[pid], **kw)
    assert tok.calls and tok.calls[0]["padding"] == "longest" and tok.calls[0]["truncation"] is True
    assert tok.calls[0]["max_length"] == 256 and tok.calls[0]["return_tensors"] == "pt"
    assert torch.equal(a[4], b[4])
    assert rel(a[4], o["out_inputs_embeds"]) < 4e-3
    lm.get_model().bert_tokenizer = None
    with pytest.raises(RuntimeError):
        lm.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, images, prompts=["x"], **kw)


@pytest.mark.parametrize("side", ["right", "left"])
def test_batch_of_two_videos(side):
    """bsz = 2: two videos of different length with their own prompts, one call; every sample equals its single-sample
    call, the shorter one is padded on the configured side (cambrian_arch.py:1753-1822)."""
    W, o = load_fixture("pipeline_T40.npz")
    lm = loaded_lm(W, o, tokenizer_padding_side=side)
    vid = torch.from_numpy(synth.video_from_basis(o["video_basis"], o["video_coef"]))
    vid2 = vid[3:30].flip(0).contiguous()
    ids = torch.from_numpy(o["input_ids"])
    ids2 = torch.cat([ids[:, :-2], ids[:, -1:], torch.zeros(1, 1, dtype=ids.dtype)], 1)     # one token shorter + a pad
    am2 = torch.ones_like(ids2)
    am2[0, -1] = 0
    pid = [int(i) for i in o["prompt_ids"]]
    size = tuple(int(v) for v in o["image_size"])
    one = lm.prepare_inputs_labels_for_multimodal(ids, None, torch.ones_like(ids), None, ids.clone(),
                                                  [vid.unsqueeze(0), (vid + 0.01).unsqueeze(0)], image_sizes=[size],
                                                  video_indices=[None], prompts=[pid], audios=[None])
    two = lm.prepare_inputs_labels_for_multimodal(ids2, None, am2, None, ids2.clone(),
                                                  [vid2.unsqueeze(0), (vid2 + 0.01).unsqueeze(0)], image_sizes=[size],
                                                  video_indices=[None], prompts=[pid[:5] + pid[-1:]], audios=[None])
    pos = torch.arange(ids.shape[1])[None].repeat(2, 1)
    both = lm.prepare_inputs_labels_for_multimodal(torch.cat([ids, ids2]), pos, torch.cat([torch.ones_like(ids), am2]),
                                                   None, torch.cat([ids, ids2]).clone(),
                                                   [[vid, vid2], [vid + 0.01, vid2 + 0.01]], image_sizes=[size, size],
                                                   video_indices=[None, None], prompts=[pid, pid[:5] + pid[-1:]],
                                                   audios=[None, None])
    n1, n2 = one[4].shape[1], two[4].shape[1]
    assert n1 != n2 and both[4].shape[:2] == (2, max(n1, n2))
    long_, short_ = (0, 1) if n1 > n2 else (1, 0)
    singles = (one, two)
    assert torch.equal(both[4][long_], singles[long_][4][0])
    ns = min(n1, n2)
    sl = slice(max(n1, n2) - ns, None) if side == "left" else slice(0, ns)
    assert torch.equal(both[4][short_][sl], singles[short_][4][0])
    assert int(both[2][short_].sum()) == ns and bool(both[2][short_][sl].all())
    assert torch.equal(both[1][short_][sl], torch.arange(ns))
    assert len(both[8]) == 40 + 27
    assert long_ == 0 and rel(both[4][0], o["out_inputs_embeds"][0]) < 4e-3


def test_several_image_tokens_in_one_sample_fail_like_the_reference():
    """The reference consumes one entry of its per-video feature list per <image> token (cur_image_idx, :1457-1495,
    :1712-1727) while the list has one entry per sample, so a sample with two <image> tokens runs off its end:
    IndexError.  The mirror raises the same exception type."""
    W, o = load_fixture("pipeline_T10_land.npz")
    lm = loaded_lm(W, o)
    vid = torch.from_numpy(synth.video_from_basis(o["video_basis"], o["video_coef"]))
    ids = torch.from_numpy(o["input_ids"])
    ids2 = torch.cat([ids, torch.tensor([[-200, 7]])], 1)
    with pytest.raises(IndexError):
        lm.prepare_inputs_labels_for_multimodal(ids2, None, None, None, None, [vid.unsqueeze(0), (vid + 0.01).unsqueeze(0)],
                                                image_sizes=[tuple(int(v) for v in o["image_size"])],
                                                video_indices=[None], prompts=[[int(i) for i in o["prompt_ids"]]],
                                                audios=[None])


@pytest.mark.parametrize("name", ["pipeline_T40.npz", "pipeline_T10_land.npz"])
def test_released_style_checkpoint_directory_to_inputs_embeds(name, tmp_path):
    """SURVEY 8(f)-4 end to end (tdc/builder.py:168-172,243-257): the fixture weights laid out as a released checkpoint
    (sharded safetensors + index, 'model.' prefix, LLM tensors mixed in) plus two HF tower directories ->
    checkpoint.load_path_weights -> the mixin -> prepare_inputs_labels_for_multimodal on the GPU == the reference's
    inputs_embeds for that video."""
    from tdc_video_amd import checkpoint as ck
    from util import write_released_style_checkpoint
    W, o = load_fixture(name)
    ckpt, sig_dir, dino_dir = write_released_style_checkpoint(W, tmp_path)
    lm = build_stub_lm(tiny_config())                                   # random init: nothing of the fixture in it yet
    before = lm.model.tdc_state_dict()["mm_projector.0.weight"].clone()
    missing, unexpected = ck.load_path_weights(lm.model, ckpt, sig_dir, dino_dir)
    assert missing == [] and unexpected == []
    assert not torch.equal(before, lm.model.tdc_state_dict()["mm_projector.0.weight"])
    with torch.no_grad():
        lm.model.embed_tokens.weight[[int(i) for i in o["used_embed_ids"]]] = torch.from_numpy(o["used_embed_rows"])
    vid = torch.from_numpy(synth.video_from_basis(o["video_basis"], o["video_coef"]))
    out = lm.prepare_inputs_labels_for_multimodal(torch.from_numpy(o["input_ids"]), None, None, None, None,
                                                  [vid.unsqueeze(0), (vid + 0.01).unsqueeze(0)],
                                                  image_sizes=[tuple(int(v) for v in o["image_size"])], video_indices=[None],
                                                  prompts=[[int(i) for i in o["prompt_ids"]]], audios=[None])
    assert rel(out[4], o["out_inputs_embeds"]) < 4e-3
    assert [list(s) for s in out[8]] == o["out_final_size"].tolist()
