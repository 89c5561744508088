"""GPU (-m gpu): every configuration BASELINE.json names, run through the HIP path at the named sizes.

  config 1  8-frame 336-px video, SigLIP + DINOv2 -> connector -> (Q-Former K=144): end to end against the oracle
            (towers cut to 2 layers at full width so the host cores finish in seconds; 336 px = 24 x 24 patches: no token
            resample, learned SigLIP positions for 576 patches, DINOv2 position table bicubic 37^2 -> 24^2)
  config 2  64-frame 336-px, bf16, H = 3584 (Qwen2-7B width): both towers at full depth against the oracle on 2 frames,
            the whole pipeline at T = 64 through its size-independent properties
  config 5  TDC-Llama3_2-3B: H = 3072, model_type "llama" (pad id 128002 in the frame budget), connector + compressor
            against the oracle on a 32-frame clip; T = 1024 with fp8 (e4m3) tower operands: determinism, batch
            invariance, token accounting
Configs 3 / 4 (8-GPU sharding, audio) live in test_hip_dist_full.py / test_hip_fullsize.py.
Tolerances: unit-norm compressed tokens atol 1e-3 in fp16 (north_star); stage outputs relative to max|ref|.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROMPT = [101] + list(range(2000, 2010)) + [102]


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tdc_oracle
    return tdc_oracle


def _rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max()) / max(1e-6, float(b.abs().max()))


def _cut_layers(sd, n_siglip, n_dino):
    """drop tower layers >= n (full width, fewer blocks: the oracle's cost is linear in depth)"""
    out = {}
    for k, v in sd.items():
        if k.startswith("vision_tower_aux_list.0.vision_tower.encoder.layers."):
            if int(k.split(".")[5]) >= n_siglip:
                continue
        if k.startswith("vision_tower_aux_list.1.vision_tower.encoder.layer."):
            if int(k.split(".")[5]) >= n_dino:
                continue
        out[k] = v
    return out


def _sd(H, K, px, dev="cuda:0", seed=0):
    import bench
    gen = torch.Generator(device=dev).manual_seed(seed)
    return bench.random_state_dict(H, K, dev, gen, siglip_px=px)


def _embed_fn(H, seed=3):
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(64, H, generator=g) * 0.02
    return lambda ids: table[torch.as_tensor(ids, dtype=torch.long) % 64]


_ORACLE_CACHE = {}


# ------------------------------------------------------------------------------------------------------------ config 1
@pytest.mark.parametrize("mns,expect_qformer", [(24, False), (2, True)])
def test_config1_8frames_336px_end_to_end_vs_oracle(mns, expect_qformer):
    """8 frames at 336 px through a1-a19 against the oracle (fp16).  max_num_segments = 24 is the released setting: 8 <= 25
    frames are all static (cambrian_arch.py:808-812, D4) and no Q-Former runs; with max_num_segments = 2 the same clip is
    segmented at its two scene cuts and the K = 144 Q-Former compresses the non-key frames."""
    import bench
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    orc = _oracle()
    H, K, T, px = 3584, 144, 8, 336
    sd = {k: v.float().cpu() for k, v in _cut_layers(_sd(H, K, px), 2, 2).items()}
    cfg = bench.model_cfg(H, K, T)
    cfg.update(max_num_segments=mns, siglip_heads=16, dino_heads=24, qformer_heads=12)
    enc = VideoEncoder(sd, cfg, dtype=torch.float16, device="cuda:0", siglip_heads=16, dino_heads=24, qformer_heads=12)
    vs = bench.synth_video(0, T, px, "cuda:0", torch.float16, scene_len=3)
    vd = bench.synth_video(0, T, px, "cuda:0", torch.float16, seed=4321, scene_len=3)
    ids = torch.tensor([[1, 2, 3, -200, 4, 5]])
    keep = {}
    got = enc.encode_video(vs, vd, (336, 336), budget_text_len=ids.shape[1], n_text_tokens=ids.shape[1] - 1,
                           prompt_ids=PROMPT, keep=keep)
    W = dict(sd)
    W["embed_tokens_fn"] = _embed_fn(H)
    with torch.no_grad():
        r = orc.encode_video(W, cfg, vs.float().cpu(), vd.float().cpu(), (336, 336), ids, torch.tensor(PROMPT))
    assert keep["seg_indices"] == [int(i) for i in r["seg_indices"]]
    assert keep["selected"] == [int(i) for i in r["selected"]]
    assert [list(s) for s in keep["final_size"]] == [list(s) for s in r["final_size"]]
    assert _rel(keep["siglip_feat"][:, :1152].reshape(T, 576, 1152), r["siglip_feat"]) < 4e-3
    assert _rel(keep["dino_feat"][:, :1536].reshape(T, 576, 1536), r["dino_feat"]) < 4e-3
    want = r["visual_tokens"]
    assert tuple(got.shape) == tuple(want.shape)
    plan = keep["plan"]
    assert (len(plan["comp_frames"]) > 0) == expect_qformer
    comp_rows = [i for i, e in enumerate(plan["src"]) if e[0] == "c"]
    stat_rows = [i for i, e in enumerate(plan["src"]) if e[0] != "c"]
    assert _rel(got[stat_rows], want[stat_rows]) < 4e-3
    if expect_qformer:
        assert keep["seg_indices"] == [2, 5]                                    # the two scene cuts of the synthetic clip
        err = float((got[comp_rows].float().cpu() - want[comp_rows]).abs().max())
        print("config 1: compressed-token max abs err %.3e" % err)
        assert err < 1e-3, err


_FULL_DEPTH_REF = {}      # px -> the oracle's run of the 8-frame clip at full tower depth (minutes of host time: shared by the cases)


def _full_depth_reference(px_s, px_d, T, H, K, mns):
    key = (px_s, px_d, T, H, K, mns)
    if key not in _FULL_DEPTH_REF:
        import bench
        orc = _oracle()
        sd = {k: v.float().cpu() for k, v in _sd(H, K, px_s).items()}
        cfg = bench.model_cfg(H, K, T)
        cfg.update(max_num_segments=mns, siglip_heads=16, dino_heads=24, qformer_heads=12)
        vs = bench.synth_video(0, T, px_s, "cuda:0", torch.float16, scene_len=3)
        vd = bench.synth_video(0, T, px_d, "cuda:0", torch.float16, seed=4321, scene_len=3)
        ids = torch.tensor([[1, 2, 3, -200, 4, 5]])
        W = dict(sd)
        W["embed_tokens_fn"] = _embed_fn(H)
        with torch.no_grad():
            r = orc.encode_video(W, cfg, vs.float().cpu(), vd.float().cpu(), (px_s, px_s), ids, torch.tensor(PROMPT))
            sims_ref = orc.adjacent_cosine(r["dino_feat"])
        _FULL_DEPTH_REF[key] = dict(sd=sd, cfg=cfg, vs=vs, vd=vd, ids=ids, r=r, sims_ref=sims_ref)
    return _FULL_DEPTH_REF[key]


# static rows element by element: |err| <= atol + rtol |ref| with rtol = 2^-10 (fp16 towers: half an fp16 ulp of the value) or 2^-9
# (bf16 tower operands).  Measured (round 5, profiles/r05_config1_full_depth_test.log): the part of |err| the relative term does
# not cover is 1.50e-3 with fp16 towers and 3.5-3.9e-3 with bf16 operands - on elements of small magnitude, i.e. accumulated
# 16-bit error of the towers and the fp16 connector, not output rounding; the north_star's literal 1e-3 atol holds for the
# unit-norm compressed tokens only (1.2e-4 measured), and these constants (<= 1.5 x measured) are what is claimed for static rows
STATIC_ATOL = {torch.float16: 2.2e-3, torch.bfloat16: 5.5e-3}


@pytest.mark.parametrize("tower_dtype,res_dtype,px,K", [               # all at the BENCH's own geometry: 384 / 378 px, 27 x 27 -> 24 x 24
    (torch.float16, None, 384, 144), (torch.bfloat16, None, 384, 144),  # fp32 residual stream (rounds 2-3)
    (torch.float16, torch.float16, 384, 144),                           # the reference's own arithmetic: fp16 throughout
    (torch.bfloat16, torch.float16, 384, 144),                          # the bench's type
    (torch.float16, torch.float16, 384, 16),                            # ... and both at K = 16, the released checkpoints'
    (torch.bfloat16, torch.float16, 384, 16),                           #     context_token_num (only S10 differs: same tower reference)
])   # (336 px - no 27 -> 24 resample, round 3's geometry - was measured for all of these in rounds 3-4: DESIGN.md section 2,
     #  profiles/r04_config1_full_depth_test.log; one geometry = one run of the fp32 oracle = a minute of the suite)
def test_config1_full_depth_end_to_end_vs_oracle(tower_dtype, res_dtype, px, K):
    """BASELINE config 1 with NOTHING cut: 8 frames, pixels -> 27-layer SigLIP / 40-layer DINOv2 -> connector -> Q-Former
    (K = 144, max_num_segments = 2 so that the 8 frames are segmented and compressed) -> emitted tokens, against
    oracle.encode_video (cambrian_arch.py:946-966,1653-1667).  px = 336: the configuration's own size (24 x 24 patches, no
    token resample); px = 384: the BENCH's geometry (SigLIP at 384 px, DINOv2 at 378 px, 27 x 27 patches bilinearly resampled
    to 24 x 24, DINOv2 positions bicubic 37^2 -> 27^2) in the bench's type - the composition the headline number is quoted on.
    Checked: segment / frame selection bit-exact; the adjacent-frame similarities of the 40-layer DINOv2 features and the
    margin between the ranks that decide the selection; compressed (unit-norm) tokens <= 1e-3 abs - the north_star's tolerance,
    through the whole composition; static rows (projector outputs of magnitude up to 4.1, where ONE fp16 ulp is 3.9e-3, so a
    literal 1e-3 atol is not something 16-bit outputs can meet - the reference's own fp16 inference included) in the form that
    IS claimable, element by element: |err| <= 1e-3 + 2^-10 |ref| with fp16 towers (half an fp16 ulp of the value on top of the
    north_star's atol) was the hoped-for form and is NOT met: measured 1.5e-3 + 2^-10 |ref| with fp16 towers, 3.7e-3 + 2^-9 |ref|
    in the bench's type (STATIC_ATOL above holds what is asserted) - and <= 7e-4 / 1.5e-3 of max|ref| as one number.
    tower_dtype = bfloat16 with res_dtype = float16 is the bench's type (bf16 GEMM operands in the towers, their residual
    stream in fp16, fp16 connector / Q-Former); the tower features themselves carry the bf16 error.  Bounds are <= 1.5 x what
    was measured (profiles/r05_config1_full_depth_test.log), so that a regression shows."""
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    orc = _oracle()
    H, T, mns = 3584, 8, 2
    px_d = px - 6 if px == 384 else px
    ref = _full_depth_reference(px, px_d, T, H, 144, mns)
    sd, cfg, vs, vd, ids, r, sims_ref = (ref[k] for k in ("sd", "cfg", "vs", "vd", "ids", "r", "sims_ref"))
    if K != 144:
        # only a11-a19 depend on K: the oracle's compressor on the cached full-depth stages (cambrian_arch.py:1507-1709)
        cfg = dict(cfg, context_token_num=K)
        if ("vis", K) not in ref:
            with torch.no_grad():
                frames, _ = orc.unpad_newline(r["mm_proj"], [(px, px)] * T, sd["image_newline"])
                ref[("vis", K)] = orc.tdc_compress(torch.stack(frames), r["seg_indices"], torch.tensor(PROMPT), sd, K, 12,
                                                   cfg["tokenizer_model_max_length"] - 16 - (ids.shape[1] - 1))
        r = dict(r, visual_tokens=ref[("vis", K)])
    enc = VideoEncoder(sd, cfg, dtype=torch.float16, tower_dtype=tower_dtype, device="cuda:0", siglip_heads=16, dino_heads=24,
                       qformer_heads=12, tower_res_dtype=res_dtype)
    assert len(enc.towers["siglip"].layers) == 27 and len(enc.towers["dino"].layers) == 40
    # raw tower features, of max|ref|: fp16 operands 3.8e-3 over an fp32 residual stream (measured 8.3e-4 / 2.5e-3), 6e-3 over an fp16
    # one (3.1e-3 / 4.1e-3: 80 residual adds rounded to 11 bits each - the reference's own fp16 arithmetic); bf16 operands
    # 3e-2 either way (6.9e-3 / 2.3e-2)
    tol_tower = 3e-2 if tower_dtype == torch.bfloat16 else 6e-3 if res_dtype is not None else 3.8e-3
    keep = {}
    got = enc.encode_video(vs, vd, (px, px), budget_text_len=ids.shape[1], n_text_tokens=ids.shape[1] - 1,
                           prompt_ids=PROMPT, keep=keep)
    sims_hip = enc.sims_tensor(keep["dino_feat"], T).cpu()
    # a5 at 40 layers: the similarities themselves and the margin of the ranking that selects the boundaries
    sim_err = float((sims_hip - sims_ref).abs().max())
    srt = torch.sort(sims_ref)[0]
    margin = float(srt[mns] - srt[mns - 1])
    es = _rel(keep["siglip_feat"][:, :1152].reshape(T, 576, 1152), r["siglip_feat"])
    ed = _rel(keep["dino_feat"][:, :1536].reshape(T, 576, 1536), r["dino_feat"])
    plan = keep["plan"]
    comp_rows = [i for i, e in enumerate(plan["src"]) if e[0] == "c"]
    stat_rows = [i for i, e in enumerate(plan["src"]) if e[0] != "c"]
    want = r["visual_tokens"]
    assert tuple(got.shape) == tuple(want.shape)
    e_stat = _rel(got[stat_rows], want[stat_rows])
    a_stat = float((got[stat_rows].float().cpu() - want[stat_rows]).abs().max())
    e_comp = float((got[comp_rows].float().cpu() - want[comp_rows]).abs().max())
    # static rows element by element: the part of |err| that a relative term rtol |ref| does not cover (rtol = half an fp16 ulp,
    # one fp16 ulp)
    d_stat = (got[stat_rows].float().cpu() - want[stat_rows]).abs()
    x10 = float((d_stat - 2.0 ** -10 * want[stat_rows].abs()).max())
    x9 = float((d_stat - 2.0 ** -9 * want[stat_rows].abs()).max())
    print("static rows, max over elements of |err| - 2^-10 |ref|: %.3e, of |err| - 2^-9 |ref|: %.3e" % (x10, x9))
    print("config 1 full depth K=%d @%d/%d px, towers %s (residual stream %s) / rest fp16: towers siglip %.3e dino %.3e (of max|ref|); "
          "similarities max abs err %.3e, ranking margin %.3e; static rows %.3e of max|ref| = %.3e abs (max|ref| %.3f); compressed "
          "tokens max abs err %.3e" % (K, px, px_d, tower_dtype, res_dtype or "fp32", es, ed, sim_err, margin, e_stat, a_stat,
                                       float(want[stat_rows].abs().max()), e_comp))
    assert keep["seg_indices"] == [int(i) for i in r["seg_indices"]] == [2, 5]
    assert keep["selected"] == [int(i) for i in r["selected"]]
    assert [list(s) for s in keep["final_size"]] == [list(s) for s in r["final_size"]]
    assert sim_err < (1e-4 if tower_dtype == torch.float16 else 2e-3) and sim_err < 0.01 * margin, (sim_err, margin)
    assert es < tol_tower and ed < tol_tower, (es, ed)
    assert len(comp_rows) > 0 and e_stat < (1.5e-3 if tower_dtype == torch.bfloat16 else 7e-4), e_stat
    assert (x9 if tower_dtype == torch.bfloat16 else x10) < STATIC_ATOL[tower_dtype], (x10, x9)   # |err| <= atol + rtol |ref| per element
    assert e_comp < 1e-3, e_comp


# ------------------------------------------------------------------------------------------------------------ config 2
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4.5e-3), (torch.bfloat16, 3e-2)])   # measured 2.9e-3 / 2.1e-2 (DINOv2)
def test_config2_336px_towers_full_depth_vs_oracle(dtype, tol):
    """SigLIP-so400m and DINOv2-giant at full depth and width on 336-px inputs, 2 frames, against the fp32 oracle: 24 x 24
    patches, so the token grid is taken as is (siglip_encoder.py:43-69 / dino_encoder.py:81-107 are the identity) and the
    DINOv2 position table is bicubic-resampled 37^2 -> 24^2."""
    import bench
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    orc = _oracle()
    H, K, px = 3584, 144, 336
    sd = {k: v.float().cpu() for k, v in _sd(H, K, px).items()}
    enc = VideoEncoder(sd, bench.model_cfg(H, K, 2), dtype=dtype, device="cuda:0", tower_batch=2)
    g = torch.Generator().manual_seed(5)
    xs = torch.rand(2, 3, px, px, generator=g) * 2 - 1
    xd = torch.rand(2, 3, px, px, generator=g) * 2 - 1
    Ws = {k[len("vision_tower_aux_list.0.vision_tower."):]: v for k, v in sd.items() if k.startswith("vision_tower_aux_list.0.")}
    Wd = {k[len("vision_tower_aux_list.1.vision_tower."):]: v for k, v in sd.items() if k.startswith("vision_tower_aux_list.1.")}
    assert Ws["embeddings.position_embedding.weight"].shape[0] == 576
    if "cfg2_towers" not in _ORACLE_CACHE:     # the same weights and inputs in both cases: one run of the (slow) fp32 oracle
        with torch.no_grad():
            _ORACLE_CACHE["cfg2_towers"] = (orc.siglip_tower(xs, Ws, 16)[0], orc.dino_tower(xd, Wd, 24)[0])
    ref_s, ref_d = _ORACLE_CACHE["cfg2_towers"]
    got_s = enc.tower("siglip", xs.cuda())[:, :1152].reshape(2, 576, 1152)
    got_d = enc.tower("dino", xd.cuda())[:, :1536].reshape(2, 576, 1536)
    es, ed = _rel(got_s, ref_s), _rel(got_d, ref_d)
    print("336-px full-depth towers %s: siglip %.3e dino %.3e" % (dtype, es, ed))
    assert es < tol and ed < tol, (es, ed)


@pytest.fixture(scope="module")
def cfg2():
    import bench
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    H, K, T, px = 3584, 144, 64, 336
    enc = VideoEncoder(_sd(H, K, px), bench.model_cfg(H, K, T), dtype=torch.bfloat16, device="cuda:0", tower_batch=64)
    torch.cuda.empty_cache()
    vs = bench.synth_video(0, T, px, "cuda:0", torch.bfloat16, scene_len=5)
    return enc, vs, T, K, H


def test_config2_T64_336px_bf16_pipeline_properties(cfg2):
    """BASELINE config 2 itself: 64 frames, 336 px, bf16, H = 3584, K = 144, full-depth towers."""
    enc, vs, T, K, H = cfg2

    def run(keep=None):
        return enc.encode_video(vs, vs, (336, 336), budget_text_len=64, n_text_tokens=64, prompt_ids=PROMPT, keep=keep,
                                frame_cap=T)
    ka, kb = {}, {}
    enc.tower_batch = 64
    a = run(ka)
    enc.tower_batch = 24                      # ragged last batch
    b = run(kb)
    enc.tower_batch = 64
    assert ka["seg_indices"] == kb["seg_indices"] and len(ka["seg_indices"]) == 24
    assert torch.equal(ka["dino_feat"], kb["dino_feat"]) and torch.equal(ka["siglip_feat"], kb["siglip_feat"])
    assert torch.equal(a, b) and torch.equal(a, run())                         # batch invariant, reproducible
    plan, N = ka["plan"], 156
    n_static, n_comp = len(plan["chunks"]), len(plan["comp_frames"])
    assert n_static + n_comp == T and n_comp > 0
    assert a.shape == (n_static * (N + 1) + n_comp * (K + 1), H) and torch.isfinite(a.float()).all()
    comp = ka["compressed"][:, :H].float()
    assert (comp.norm(dim=-1) - 1.0).abs().max().item() < 4e-3                 # bf16 rows of unit L2 norm
    X = ka["X"]
    for j in range(0, len(plan["src"]), 89):
        e = plan["src"][j]
        want = enc.c.frame_seg[0, :H] if e[0] == "s" else X[e[1] * N + e[2], :H] if e[0] == "f" else ka["compressed"][e[1] * K + e[2], :H]
        assert torch.equal(a[j], want)


# ------------------------------------------------------------------------------------------------------------ config 5
@pytest.mark.parametrize("K,mns", [(144, 24), (16, 3)])     # K = 16: the released checkpoints' context_token_num (train_video_qwen.sh:51-52,63)
def test_config5_llama_H3072_connector_compressor_vs_oracle(K, mns):
    """S4-S10 at the Llama-3.2-3B width (H = 3072) on a 32-frame clip of tower features vs the oracle (fp16); K = 144 with the
    released max_num_segments = 24 (7 compressed frames), K = 16 with 3 segments (> 20 compressed frames)."""
    import bench
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    from tdc_video_amd import segment as seg
    from tdc_video_amd.weights import pad64
    orc = _oracle()
    H, T = 3072, 32
    sd = {k: v.float().cpu() for k, v in _sd(H, K, 384).items() if not k.startswith("vision_tower_aux_list")}
    cfg = bench.model_cfg(H, K, T)
    cfg["model_type"] = "llama"
    cfg["max_num_segments"] = mns
    enc = VideoEncoder(sd, cfg, dtype=torch.float16, device="cuda:0")
    g = torch.Generator().manual_seed(9)
    sig = torch.randn(T, 576, 1152, generator=g).half().float()
    din = torch.randn(T, 576, 1536, generator=g).half().float()
    with torch.no_grad():
        aux = [orc.mm_projector_aux(sig, sd, 0), orc.mm_projector_aux(din, sd, 1)]
        q, _ = orc.sva(aux, sd["vision_query"][0], [(384, 384)] * T, sd, 12)
        feat = orc.mm_projector(q, sd)
        frames, _ = orc.unpad_newline(feat, [(384, 384)] * T, sd["image_newline"])
        segi = orc.select_segments(orc.adjacent_cosine(din), mns)
        want = orc.tdc_compress(torch.stack(frames), segi, torch.tensor(PROMPT), sd, K, 12, 10 ** 9)

    def pad(x, D):
        buf = torch.zeros(x.shape[0] * x.shape[1], pad64(D), dtype=torch.float16, device="cuda:0")
        buf[:, :D] = x.reshape(-1, D).half().cuda()
        return buf
    keep = {}
    X, _ = enc.connector(pad(sig, 1152), pad(din, 1536), T, [(384, 384)] * T, keep)
    assert seg.select_segments(enc.sims_tensor(pad(din, 1536), T).tolist(), mns) == [int(i) for i in segi]
    got = enc.compress(X, T, X.shape[0] // T, [int(i) for i in segi], PROMPT, 10 ** 9, keep=keep)
    assert tuple(got.shape) == tuple(want.shape) and got.shape[1] == H
    plan = keep["plan"]
    comp_rows = [i for i, e in enumerate(plan["src"]) if e[0] == "c"]
    stat_rows = [i for i, e in enumerate(plan["src"]) if e[0] == "f"]
    err_c = float((got[comp_rows].float().cpu() - want[comp_rows]).abs().max())
    err_s = _rel(got[stat_rows], want[stat_rows])
    assert len(plan["comp_frames"]) >= (7 if mns == 24 else 20)
    print("H=3072 K=%d (%d compressed frames): compressed-token max abs err %.3e, static rows rel %.3e"
          % (K, len(plan["comp_frames"]), err_c, err_s))
    assert err_c < 1e-3 and err_s < 8e-4, (err_c, err_s)        # measured 1.0e-4 / 5.3e-4


# per level: (compressed tokens max abs, static rows of max|bf16|, similarities max abs) at <= 1.5 x what the MI355X measured
# (profiles/r06_config5_fp8_contract.log: level 1 1.22e-3 / 9.8e-3 / 6.9e-3, level 2 1.71e-3 / 1.33e-2 / 1.32e-2, level 3 1.71e-3 /
# 1.39e-2 / 1.30e-2; tower features move by 5.8-8.8 % (SigLIP) / 19-24 % (DINOv2) RMS - e4m3's 3 mantissa bits through 27 / 40
# layers of random-init weights - and the unit-norm context tokens behind the Q-Former by 1.2-1.7e-3)
FP8_CONTRACT = {1: (1.9e-3, 1.5e-2, 1.05e-2), 2: (2.6e-3, 2.0e-2, 2.0e-2), 3: (2.6e-3, 2.1e-2, 2.0e-2)}


@pytest.mark.parametrize("level", [1, 2, 3])
def test_config5_fp8_contract_vs_bf16(level):
    """BASELINE config 5's arithmetic at the final code - e4m3 tower operands (level 1: qkv / fc1; 2: all four tower GEMMs; 3: fc1
    writes the e4m3 hidden itself) OVER THE fp16 RESIDUAL STREAM (round 6: the fp8 levels compose with tdc_vit_model.res_dtype_p1) -
    held to a measured contract against the bf16-operand path on the same weights and pixels: TDC-Llama3_2-3B width (H = 3072, pad id
    128002), both towers at full depth, 8 frames, max_num_segments = 2 so that the Q-Former runs.  e4m3 has 3 mantissa bits: this is a
    THROUGHPUT mode (bench.py says so in its line), its outputs are not held to the oracle's tolerances but to: the same segment
    selection on a video with real scene cuts; compressed (unit-norm) context tokens, static rows and adjacent-frame similarities
    within FP8_CONTRACT of the bf16 path's; tdc_vit_fwd == the per-kernel sequence bit for bit; bitwise determinism."""
    import bench
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    H, K, T, mns = 3072, 144, 8, 2
    cfg = bench.model_cfg(H, K, T)
    cfg.update(model_type="llama", max_num_segments=mns)
    if "sd" not in _FP8_CACHE:
        sd = _sd(H, K, 384)
        vs = bench.synth_video(0, T, 384, "cuda:0", torch.bfloat16, scene_len=3)
        vd = bench.synth_video(0, T, 378, "cuda:0", torch.bfloat16, seed=4321, scene_len=3)
        enc = VideoEncoder(sd, cfg, dtype=torch.bfloat16, device="cuda:0", tower_res_dtype=torch.float16)
        keep = {}
        base = enc.encode_video(vs, vd, (384, 384), budget_text_len=64, n_text_tokens=64, prompt_ids=PROMPT, keep=keep)
        _FP8_CACHE.update(sd=sd, vs=vs, vd=vd, base=base.float(), keep=keep, sims=enc.sims_tensor(keep["dino_feat"], T).cpu())
        del enc
    sd, vs, vd, base, kb, sims_b = (_FP8_CACHE[k] for k in ("sd", "vs", "vd", "base", "keep", "sims"))
    enc = VideoEncoder(sd, cfg, dtype=torch.bfloat16, device="cuda:0", fp8_towers=level, tower_res_dtype=torch.float16)
    assert all(int(t.fp8) == level for t in enc.towers.values()) and enc.tower_res_dtype == torch.float16
    keep = {}
    got = enc.encode_video(vs, vd, (384, 384), budget_text_len=64, n_text_tokens=64, prompt_ids=PROMPT, keep=keep)
    assert torch.equal(got, enc.encode_video(vs, vd, (384, 384), budget_text_len=64, n_text_tokens=64, prompt_ids=PROMPT))
    enc.native_towers = False                      # the same launches issued one by one from Python
    assert torch.equal(enc.tower("dino", vd), keep["dino_feat"]) and torch.equal(enc.tower("siglip", vs), keep["siglip_feat"])
    enc.native_towers = True
    assert keep["seg_indices"] == kb["seg_indices"] == [2, 5] and keep["selected"] == kb["selected"]
    plan = keep["plan"]
    comp_rows = [i for i, e in enumerate(plan["src"]) if e[0] == "c"]
    stat_rows = [i for i, e in enumerate(plan["src"]) if e[0] == "f"]
    assert got.shape == base.shape and len(comp_rows) == 5 * K
    e_comp = float((got[comp_rows].float() - base[comp_rows]).abs().max())
    e_stat = float((got[stat_rows].float() - base[stat_rows]).abs().max() / base[stat_rows].abs().max())
    e_sim = float((enc.sims_tensor(keep["dino_feat"], T).cpu() - sims_b).abs().max())
    srt = torch.sort(sims_b)[0]
    rms = lambda a, b: float(((a.float() - b.float()).pow(2).mean().sqrt() / b.float().pow(2).mean().sqrt()))   # noqa: E731
    print("config 5 fp8 level %d over the fp16 stream vs bf16: compressed tokens max abs %.3e, static rows %.3e of max, similarities "
          "max abs %.3e (ranking margin %.3f), towers rel RMS siglip %.3e dino %.3e"
          % (level, e_comp, e_stat, e_sim, float(srt[mns] - srt[mns - 1]), rms(keep["siglip_feat"], kb["siglip_feat"]),
             rms(keep["dino_feat"], kb["dino_feat"])))
    c, s_, m_ = FP8_CONTRACT[level]
    assert e_comp < c and e_stat < s_ and e_sim < m_, (e_comp, e_stat, e_sim)
    assert (got[comp_rows].float().norm(dim=-1) - 1.0).abs().max().item() < 4e-3


_FP8_CACHE = {}


@pytest.mark.parametrize("level", [1])
def test_config5_T1024_fp8_towers_properties(level):
    """BASELINE config 5 on one GPU: 1024 frames, H = 3072, e4m3 operands for the towers' LayerNorm-fed GEMMs (level 1) over the
    fp16 residual stream, full depth: bitwise batch invariance and determinism, token accounting, unit-norm context rows,
    verbatim static rows."""
    import bench
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    H, K, T = 3072, 144, 1024
    cfg = bench.model_cfg(H, K, T)
    cfg["model_type"] = "llama"
    enc = VideoEncoder(_sd(H, K, 384), cfg, dtype=torch.bfloat16, device="cuda:0", tower_batch=512, fp8_towers=level,
                       tower_res_dtype=torch.float16)
    assert all(int(t.fp8) == level for t in enc.towers.values())
    torch.cuda.empty_cache()
    vs = bench.synth_video(0, T, 384, "cuda:0", torch.bfloat16)
    vd = bench.synth_video(0, T, 378, "cuda:0", torch.bfloat16, seed=4321)

    def run(keep=None):
        return enc.encode_video(vs, vd, (384, 384), budget_text_len=64, n_text_tokens=64, prompt_ids=PROMPT, keep=keep,
                                frame_cap=T)
    ka, kb = {}, {}
    a = run(ka)
    enc.tower_batch = 200                     # 5 full batches + a 24-frame one
    b = run(kb)
    assert ka["seg_indices"] == kb["seg_indices"] and len(ka["seg_indices"]) == 24
    assert torch.equal(ka["dino_feat"], kb["dino_feat"]) and torch.equal(a, b)
    plan, N = ka["plan"], 156
    n_static, n_comp = len(plan["chunks"]), len(plan["comp_frames"])
    assert n_static + n_comp == T and a.shape == (n_static * (N + 1) + n_comp * (K + 1), H)
    assert torch.isfinite(a.float()).all()
    assert (ka["compressed"][:, :H].float().norm(dim=-1) - 1.0).abs().max().item() < 4e-3
    del enc
    torch.cuda.empty_cache()


def test_llama_pad_id_in_the_frame_budget():
    """cambrian_arch.py:753-757: the text length of the frame budget ends at the first pad id - 128002 for a model_type
    containing 'llama', 151643 otherwise - and the budget then caps the frames of a long video (a1) on the device path."""
    from test_host_logic import build_stub_lm, tiny_config
    import synth
    from util import load_fixture
    W, o = load_fixture("pipeline_T40.npz")
    ids = torch.from_numpy(o["input_ids"])
    pad_l, pad_q = 128002, 151643
    padded_l = torch.cat([ids, torch.full((1, 7), pad_l)], 1)
    padded_q = torch.cat([ids, torch.full((1, 7), pad_q)], 1)
    for mt, own, other in (("llama", padded_l, padded_q), ("qwen2", padded_q, padded_l)):
        lm = build_stub_lm(tiny_config(model_type=mt))
        assert lm._budget_text_len(own[0]) == ids.shape[1]
        assert lm._budget_text_len(other[0]) == ids.shape[1] + 7
    # end to end on the device: a budget of 12 frames caps the 40-frame fixture video
    lm = build_stub_lm(tiny_config(model_type="llama", tokenizer_model_max_length=ids.shape[1] + 16 + 12 * ((144 + 4 * 7) // 8)))
    m = lm.model
    m.load_state_dict({k: v for k, v in W.items() if not k.startswith("vision_tower_aux_list")}, strict=False)
    for i, t in enumerate(m.vision_tower_aux_list):
        pre = "vision_tower_aux_list.%d.vision_tower." % i
        t.load_model(state_dict={k[len(pre):]: v for k, v in W.items() if k.startswith(pre)})
    vid = torch.from_numpy(synth.video_from_basis(o["video_basis"], o["video_coef"]))
    am = torch.cat([torch.ones_like(ids), torch.zeros(1, 7, dtype=ids.dtype)], 1)
    out = lm.prepare_inputs_labels_for_multimodal(padded_l, None, am, None, None, [vid.unsqueeze(0), (vid + 0.01).unsqueeze(0)],
                                                  image_sizes=[tuple(int(v) for v in o["image_size"])],
                                                  video_indices=[None], prompts=[[int(i) for i in o["prompt_ids"]]],
                                                  audios=[None])
    assert lm.get_max_num_frames(padded_l[0]) == 12
    assert len(out[8]) == 12                                 # final_size: one entry per frame that survived the cap


# ------------------------------------------------------------------------------------------------------------ a5 ties
def test_a5_exact_ties_and_near_ties_of_the_similarity_ranking():
    """The reference ranks the adjacent-frame similarities with torch.argsort (cambrian_arch.py:849), which promises no
    order among equal values; this path ranks (value, index) - stable, ties go to the LOWEST index - and its fp32 similarity
    kernel reduces in a fixed order, so equal frame pairs give bit-equal similarities and the selection is reproducible.
    Documented consequence: with exact or sub-ulp ties the selected segment boundaries are a property of this
    implementation, not of the reference; parity of `seg_indices` is claimed (and tested) where the 24th and 25th smallest
    similarities differ by more than fp32 rounding."""
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd import ops, segment as seg
    orc = _oracle()
    g = torch.Generator().manual_seed(0)
    P, D, T = 4, 64, 30
    A = torch.randn(P, D, generator=g).half()
    B = (A.float() + 0.5 * torch.randn(P, D, generator=g)).half()
    frames = torch.stack([A if t % 2 == 0 else B for t in range(T)])            # A B A B ...: 29 identical pairs
    feat = frames.reshape(T * P, D).cuda().contiguous()
    sims = ops.frame_cossim(feat, T, P * D).tolist()
    assert len(set(sims)) == 1, "equal pairs (in either order) must give bit-equal similarities"
    assert seg.select_segments(sims, 24) == list(range(24))                     # exact ties: lowest indices, in order
    # one pair made slightly LESS similar, late in the video: it is ranked first, the rest stay lowest-index-first
    frames2 = frames.clone()
    frames2[27] = (frames2[27].float() + 0.05 * torch.randn(P, D, generator=g)).half()
    sims2 = ops.frame_cossim(frames2.reshape(T * P, D).cuda().contiguous(), T, P * D).tolist()
    pick = seg.select_segments(sims2, 24)
    low = sorted(range(T - 1), key=lambda i: (sims2[i], i))[:2]
    assert set(low) == {26, 27} and 26 in pick and 27 in pick and pick[:22] == list(range(22))
    assert ops.frame_cossim(frames2.reshape(T * P, D).cuda().contiguous(), T, P * D).tolist() == sims2   # reproducible
    # where the gap is real the oracle agrees on the index set; inside the tie it may not - that is the documented limit
    ref = orc.adjacent_cosine(frames2.float().reshape(T, P, D))
    assert {26, 27} <= set(int(i) for i in orc.select_segments(ref, 24))
    # near-ties 1 ulp apart are ranked by value
    s = [0.5] * 10
    s[7] = float(np.nextafter(np.float32(0.5), np.float32(0.0)))
    assert seg.select_segments(s, 3) == [0, 1, 7]
