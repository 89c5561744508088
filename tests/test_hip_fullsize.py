"""GPU, BASELINE.json's full sizes (SigLIP-so400m + DINOv2-giant architectures, H=3584, K=144 / 16): the oracle is far
too slow there, so parity is carried by size-independent properties of the path (run with -m gpu):
  * batch invariance: frames are independent through S1-S9, so tower batch size must not change a single bit;
  * run-to-run determinism (bitwise), incl. the similarity ranking;
  * emitted layout: static tokens are verbatim mm_projector rows, separators are frame_seg, compressed rows have
    unit L2 norm, and the token count obeys SURVEY appendix B;
  * the Q-Former is per-frame independent given its chunk's key frame: compressing a subset of frames of a chunk
    gives the same rows;
  * sharded (world 1, RCCL) == serial."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def full():
    import bench
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    H, K, T = 3584, 144, 40
    cfg = bench.model_cfg(H, K, T)
    sd = bench.random_state_dict(H, K, dev, gen)
    enc = VideoEncoder(sd, cfg, dtype=torch.bfloat16, device=dev, tower_batch=16)
    del sd
    torch.cuda.empty_cache()
    vs = bench.synth_video(0, T, 384, dev, torch.bfloat16, scene_len=5)
    vd = bench.synth_video(0, T, 378, dev, torch.bfloat16, seed=4321, scene_len=5)
    return enc, vs, vd, T, K, H


def run(enc, vs, vd, keep=None, **kw):
    return enc.encode_video(vs, vd, (384, 384), budget_text_len=64, n_text_tokens=64,
                            prompt_ids=[101] + list(range(2000, 2010)) + [102], keep=keep, **kw)


def test_batch_invariance_and_determinism(full):
    enc, vs, vd, T, K, H = full
    keep_a, keep_b = {}, {}
    enc.tower_batch = 16
    a = run(enc, vs, vd, keep_a)
    enc.tower_batch = 7            # ragged last batch
    b = run(enc, vs, vd, keep_b)
    assert keep_a["seg_indices"] == keep_b["seg_indices"]
    assert torch.equal(keep_a["dino_feat"], keep_b["dino_feat"])
    assert torch.equal(keep_a["siglip_feat"], keep_b["siglip_feat"])
    assert torch.equal(a, b)
    c = run(enc, vs, vd)
    assert torch.equal(b, c)       # bitwise reproducible run to run
    enc.tower_batch = 16


def test_emitted_layout_properties(full):
    enc, vs, vd, T, K, H = full
    keep = {}
    out = run(enc, vs, vd, keep)
    plan = keep["plan"]
    N = 156
    n_static = len(plan["chunks"])
    n_comp = len(plan["comp_frames"])
    assert n_static + n_comp == T and n_comp > 0
    assert out.shape == (n_static * (N + 1) + n_comp * (K + 1), H)
    assert torch.isfinite(out.float()).all()
    seg_row = enc.c.frame_seg[0, :H]
    X = keep["X"]
    comp = keep["compressed"]
    for i, e in enumerate(plan["src"][:: 97]):
        j = i * 97
        if e[0] == "s":
            assert torch.equal(out[j], seg_row)
        elif e[0] == "f":
            assert torch.equal(out[j], X[e[1] * N + e[2], :H])
        else:
            assert torch.equal(out[j], comp[e[1] * K + e[2], :H])
    norms = comp[:, :H].float().norm(dim=-1)
    assert (norms - 1.0).abs().max().item() < 4e-3                         # bf16 rows of unit L2 norm
    # newline column: every 13th token of a static frame is image_newline
    nl = enc.c.image_newline[0, :H]
    f0 = plan["chunks"][0][0]
    assert torch.equal(X[f0 * N + 12, :H], nl) and torch.equal(X[f0 * N + 155, :H], nl)


def test_qformer_frame_independence(full):
    """a compressed frame depends only on (its own tokens, its chunk's key frame, the prompt)."""
    enc, vs, vd, T, K, H = full
    keep = {}
    run(enc, vs, vd, keep)
    plan, X = keep["plan"], keep["X"]
    N = 156
    full_comp = keep["compressed"]
    pick = [i for i in range(len(plan["comp_frames"]))][::3]
    qtable = enc.make_queries(X, N, N, plan["key_frames"])
    sub = enc.compress_frames(X, N, [plan["comp_frames"][i] for i in pick], qtable,
                              [plan["comp_chunk"][i] for i in pick], [101] + list(range(2000, 2010)) + [102])
    for n, i in enumerate(pick):
        assert torch.equal(sub[n * K:(n + 1) * K], full_comp[i * K:(i + 1) * K])


def test_sharded_world1_equals_serial_fullsize(full):
    import torch.distributed as dist
    from tdc_video_amd.dist import ShardedVideoEncoder
    enc, vs, vd, T, K, H = full
    want = run(enc, vs, vd, frame_cap=T)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got = ShardedVideoEncoder(enc, 0, 1).encode_video(vs, vd, T, (384, 384), 64,
                                                          [101] + list(range(2000, 2010)) + [102])
    finally:
        dist.destroy_process_group()
    assert torch.equal(got, want)


# ---- full-size NUMERIC parity against the oracle on bounded samples (the GPU box's host cores run the fp32 restatement:
#      2 frames through both towers, connector + compressor on a 32-frame clip; about a minute) -----------------------
def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tdc_oracle
    return tdc_oracle


def _rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max()) / max(1e-6, float(b.abs().max()))


_ORACLE_CACHE = {}


@pytest.fixture(scope="module")
def full_sd():
    import bench
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    H, K, T = 3584, 144, 40
    sd = bench.random_state_dict(H, K, dev, gen)
    return {k: v.float().cpu() for k, v in sd.items()}, bench.model_cfg(H, K, T), H, K


@pytest.mark.parametrize("fuse", ["0", "1"])
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])    # measured 2.3-2.6e-3 / 1.8-2.3e-2 (DINOv2)
def test_fullsize_towers_vs_oracle(full_sd, dtype, tol, fuse, monkeypatch):
    """SigLIP-so400m (27 x 1152, d_head 72) and DINOv2-giant (40 x 1536, SwiGLU, LayerScale) at full depth and width:
    2 frames, HIP vs the fp32 oracle.  Tolerances are relative to max|ref| of the tower output: the raw residual
    stream of a random-init 40-layer ViT is carried through 80 16-bit GEMM inputs (bf16: 8 mantissa bits)."""
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    orc = _oracle()
    sd, cfg, H, K = full_sd
    # fuse "1": pre-LayerNorms folded into the neighbouring GEMMs
    enc = VideoEncoder(sd, cfg, dtype=dtype, device="cuda:0", tower_batch=2, ln_fuse=fuse == "1")
    g = torch.Generator().manual_seed(5)
    xs = torch.rand(2, 3, 384, 384, generator=g) * 2 - 1
    xd = torch.rand(2, 3, 378, 378, generator=g) * 2 - 1
    Ws = {k[len("vision_tower_aux_list.0.vision_tower."):]: v for k, v in sd.items() if k.startswith("vision_tower_aux_list.0.")}
    Wd = {k[len("vision_tower_aux_list.1.vision_tower."):]: v for k, v in sd.items() if k.startswith("vision_tower_aux_list.1.")}
    if "towers" not in _ORACLE_CACHE:          # the same inputs in all four cases: one run of the (slow) fp32 oracle
        with torch.no_grad():
            _ORACLE_CACHE["towers"] = (orc.siglip_tower(xs, Ws, 16)[0], orc.dino_tower(xd, Wd, 24)[0])
    ref_s, ref_d = _ORACLE_CACHE["towers"]
    got_s = enc.tower("siglip", xs.cuda())[:, :1152].reshape(2, 576, 1152)
    got_d = enc.tower("dino", xd.cuda())[:, :1536].reshape(2, 576, 1536)
    es, ed = _rel(got_s, ref_s), _rel(got_d, ref_d)
    print("full-size tower rel err %s fuse=%s: siglip %.3e dino %.3e" % (dtype, fuse, es, ed))
    assert es < tol and ed < tol, (es, ed)


# K = 16 is the reference's own default and what the released checkpoints use (context_token_num,
# /root/reference/scripts/stage2/train_video_qwen.sh:51-52,63; tdc/cambrian_arch.py:1510,1633-1667); K = 144 is what BASELINE's configs name.
# At K = 16 the shapes differ in kind: S = K + Lt = 28 rows per frame in the self-attention, F * K of a few hundred rows in the
# row-mapped query GEMMs (fewer 256 x 256 tiles than CUs on the persistent kernel), a ragged last 64-row block in the fused
# cross-attention output kernel.  mns = max_num_segments: 24 leaves 7 compressed frames of the 32, 3 leaves ~26.
# size: the frames' (H, W) as the caller passes it.  (360, 640) is a 16:9 video: the SVA's window masks exclude the padding rows of
# the squared frame (tdc/cambrian_arch.py:619-669), the unpad - which reads the pair as (W, H), SURVEY D8 - keeps 6 of the 12 token
# COLUMNS (:512-544) and the Q-Former sees N = 12 x (6 + 1) = 84 encoder tokens per frame - the a7 / a10 geometry at full width
@pytest.mark.parametrize("K,mns,size", [(144, 24, (384, 384)), (16, 24, (384, 384)), (16, 3, (384, 384)), (16, 3, (360, 640))])
def test_fullsize_connector_compressor_vs_oracle(full_sd, K, mns, size):
    """S4-S10 at BASELINE sizes (C=1024, H=3584, 12-layer Q-Former) on a 32-frame clip of tower features:
    emitted tokens vs the oracle; compressed (unit-norm) rows within the north_star's 1e-3 fp16 atol."""
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    from tdc_video_amd import segment as seg
    orc = _oracle()
    sd, cfg, H, _ = full_sd
    cfg = dict(cfg, context_token_num=K, max_num_segments=mns)
    W = {k: v for k, v in sd.items() if not k.startswith("vision_tower_aux_list")}
    enc = VideoEncoder(W, cfg, dtype=torch.float16, device="cuda:0")
    T = 32
    g = torch.Generator().manual_seed(9)
    sig = torch.randn(T, 576, 1152, generator=g).half().float()
    din = torch.randn(T, 576, 1536, generator=g).half().float()
    pid = [101] + list(range(2000, 2010)) + [102]
    with torch.no_grad():
        if ("s4_s7", size) not in _ORACLE_CACHE:       # connector stages do not depend on K / mns: one run of the fp32 oracle per size
            if "aux" not in _ORACLE_CACHE:
                _ORACLE_CACHE["aux"] = [orc.mm_projector_aux(sig, W, 0), orc.mm_projector_aux(din, W, 1)]
            q, _ = orc.sva(_ORACLE_CACHE["aux"], W["vision_query"][0], [size] * T, W, 12)
            feat = orc.mm_projector(q, W)
            _ORACLE_CACHE[("s4_s7", size)] = (orc.unpad_newline(feat, [size] * T, W["image_newline"])[0], orc.adjacent_cosine(din))
        frames, sims_ref = _ORACLE_CACHE[("s4_s7", size)]
        segi = orc.select_segments(sims_ref, mns)
        want = orc.tdc_compress(torch.stack(frames), segi, torch.tensor(pid), W, K, 12, 10 ** 9)
    from tdc_video_amd.weights import pad64
    def pad(x, D):
        buf = torch.zeros(x.shape[0] * x.shape[1], pad64(D), dtype=torch.float16, device="cuda:0")
        buf[:, :D] = x.reshape(-1, D).half().cuda()
        return buf
    keep = {}
    X, fsz = enc.connector(pad(sig, 1152), pad(din, 1536), T, [size] * T, keep)
    assert X.shape[0] // T == frames[0].shape[0] == (156 if size == (384, 384) else 84), (X.shape, frames[0].shape)
    sims = enc.sims_tensor(pad(din, 1536), T).tolist()
    assert seg.select_segments(sims, mns) == [int(i) for i in segi]
    got = enc.compress(X, T, X.shape[0] // T, [int(i) for i in segi], pid, 10 ** 9, keep=keep)
    assert tuple(got.shape) == tuple(want.shape)
    plan = keep["plan"]
    comp_rows = [i for i, e in enumerate(plan["src"]) if e[0] == "c"]
    stat_rows = [i for i, e in enumerate(plan["src"]) if e[0] == "f"]
    assert len(comp_rows) == len(plan["comp_frames"]) * K and comp_rows
    assert len(plan["comp_frames"]) == T - len(plan["chunks"]) and len(plan["comp_frames"]) >= (7 if mns == 24 else 20)
    err_c = float((got[comp_rows].float().cpu() - want[comp_rows]).abs().max())
    err_s = _rel(got[stat_rows], want[stat_rows])
    print("full-size K=%d mns=%d frames %s (%d compressed frames, N = %d): compressed-token max abs err %.3e (unit-norm rows), static "
          "rows rel %.3e" % (K, mns, size, len(plan["comp_frames"]), X.shape[0] // T, err_c, err_s))
    assert err_c < 1e-3, err_c
    assert err_s < 8e-4, err_s          # measured 4.7e-4


@pytest.mark.parametrize("N", [156, 206, 84])
def test_fullwidth_compressor_K16_vs_oracle_and_kernel_sequence(full_sd, N):
    """a11-a19 at the released K = 16 and full width (H = 3584, bert-base Q-Former) for the three encoder-token counts the path
    produces: N = 156 (square frames: 12 x (12 + 1)), 206 (+ 50 audio tokens, tdc/cambrian_arch.py:1611-1614), 84 (16:9 frames
    after the unpad: 7 x (11 + 1)).  Frames are random rows at the mm_projector's scale; checked: emitted tokens vs the oracle
    (compressed rows <= 1e-3 abs), tdc_qformer_fwd == the per-kernel sequence bit for bit in every cross-attention form, and
    per-frame independence (a subset of the compressed frames gives the same rows)."""
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    orc = _oracle()
    sd, cfg, H, _ = full_sd
    K, T, mns = 16, 40, 3
    cfg = dict(cfg, context_token_num=K, max_num_segments=mns)
    W = {k: v for k, v in sd.items() if not k.startswith("vision_tower_aux_list")}
    enc = VideoEncoder(W, cfg, dtype=torch.float16, device="cuda:0")
    g = torch.Generator().manual_seed(100 + N)
    frames = torch.randn(T, N, H, generator=g).half().float()
    segi = [6, 17, 29]
    pid = [101] + list(range(2000, 2010)) + [102]
    with torch.no_grad():
        want = orc.tdc_compress(frames, torch.tensor(segi), torch.tensor(pid), W, K, 12, 10 ** 9)
    X = frames.reshape(T * N, H).half().cuda().contiguous()
    outs = {}
    for native, mode in ((False, 1), (True, 0), (False, 0), (True, 2), (False, 2), (True, 1)):     # the product's form last: `keep`
        enc.native_qformer, enc.xattn_mode = native, mode
        keep = {}
        outs[(native, mode)] = enc.compress(X, T, N, segi, pid, 10 ** 9, keep=keep)
    got = outs[(True, 1)]
    for mode in (0, 1, 2):
        assert torch.equal(outs[(True, mode)], outs[(False, mode)]), "composite != kernel sequence, xattn_mode %d" % mode
    plan = keep["plan"]
    comp_rows = [i for i, e in enumerate(plan["src"]) if e[0] == "c"]
    stat_rows = [i for i, e in enumerate(plan["src"]) if e[0] == "f"]
    F = len(plan["comp_frames"])
    assert tuple(got.shape) == tuple(want.shape) and F == 33 and len(comp_rows) == F * K      # 7 chunks of <= 8 frames
    assert torch.equal(got[stat_rows].cpu(), frames.reshape(T * N, H)[[plan["src"][i][1] * N + plan["src"][i][2] for i in stat_rows]].half())
    errs = {m: float((outs[(True, m)][comp_rows].float().cpu() - want[comp_rows]).abs().max()) for m in (0, 1, 2)}
    print("full-width K=16 N=%d: compressed-token max abs err by cross-attention form %s" % (N, errs))
    assert max(errs.values()) < 1e-3, errs
    # per-frame independence: every second compressed frame alone -> the same rows
    sub = list(range(0, F, 2))
    qtable = enc.make_queries(X, N, N, plan["key_frames"])
    part = enc.compress_frames(X, N, [plan["comp_frames"][i] for i in sub], qtable, [plan["comp_chunk"][i] for i in sub], pid)
    full_c = keep["compressed"]
    for j, i in enumerate(sub):
        assert torch.equal(part[j * K:(j + 1) * K], full_c[i * K:(i + 1) * K]), "frame %d" % i


def test_fullsize_video_plus_audio_token_accounting(full):
    """BASELINE config 4 at full sizes: T seconds of audio through BEATs (released dimensions) + a20 interleave: the
    Q-Former KV grows to N + 50 rows per frame, static frames emit N + 50 + 1 tokens, compressed frames K + 1
    (SURVEY appendix B); bitwise reproducible."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_beats import random_beats_state
    from tdc_video_amd import weights as Wt
    from tdc_video_amd.beats import BEATS_ITER3_CFG, BeatsEncoder
    enc, vs, vd, T, K, H = full
    dev = vs.device
    g = torch.Generator(device=dev).manual_seed(3)
    old = (enc.cfg.get("audio_input"), enc.c.audio_proj, enc.beats)
    try:
        enc.cfg["audio_input"] = True
        enc.c.audio_proj = Wt.make_lin(torch.randn(H, 768, device=dev, generator=g) * 0.02, torch.zeros(H, device=dev),
                                       torch.bfloat16, dev)
        enc.beats = BeatsEncoder(random_beats_state(BEATS_ITER3_CFG), BEATS_ITER3_CFG, dtype=torch.bfloat16, device=dev)
        wav = (0.1 * torch.randn(1, 16000 * T + 4321, device=dev, generator=g)).half()
        keep = {}
        a = run(enc, vs, vd, keep, audio={"audio_wav": wav, "audio_wav_mask": torch.zeros_like(wav)})
        b = run(enc, vs, vd, audio={"audio_wav": wav})
        assert torch.equal(a, b)
        plan = keep["plan"]
        N = 156
        n_static = len(plan["chunks"])
        n_comp = len(plan["comp_frames"])
        assert n_static + n_comp == T and n_comp > 0
        assert a.shape[0] == n_static * (N + 50 + 1) + n_comp * (K + 1)
        assert torch.isfinite(a.float()).all()
        # padded audio (audio_wav_mask True on the tail of the last full window and the short one): same layout, finite, and
        # different from the unpadded encode only from the first frame whose BEATs window holds padding
        mask = torch.zeros(wav.shape, dtype=torch.bool)
        mask[0, 16000 * (T - 7):] = True
        m = run(enc, vs, vd, audio={"audio_wav": wav, "audio_wav_mask": mask})
        assert m.shape == a.shape and torch.isfinite(m.float()).all() and not torch.equal(m, a)
        first_bad = int(torch.nonzero((m != a).any(1))[0])
        keys = set(int(f) for f in plan["key_frames"])
        rows_before = sum((N + 50 + 1) if f in keys else (K + 1) for f in range(30))   # frames in front of the window 30 s .. 40 s
        assert first_bad >= rows_before > 0
    finally:
        enc.cfg["audio_input"], enc.c.audio_proj, enc.beats = old
        if old[0] is None:
            enc.cfg.pop("audio_input", None)


def test_config4_T512_with_512s_of_audio_properties():
    """BASELINE config 4 at ITS OWN size, in the bench's type (bf16 tower operands, fp16 residual stream, fp16 connector /
    Q-Former): one 512-frame video + 512 s of 16 kHz audio through BEATs (released dimensions), 50 audio tokens per frame in
    the Q-Former KV (N = 206).  The oracle cannot run this size; checked are the size-independent properties: token accounting
    (SURVEY appendix B with the audio rows), bitwise run-to-run determinism, bitwise tower-batch invariance (512 frames in
    one batch == batches of 96 with a ragged tail), unit-norm compressed rows, verbatim separators."""
    import bench
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd import weights as Wt
    from tdc_video_amd.beats import BEATS_ITER3_CFG, BeatsEncoder
    from tdc_video_amd.pipeline import VideoEncoder
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_beats import random_beats_state
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    H, K, T, N = 3584, 144, 512, 156
    sd = bench.random_state_dict(H, K, dev, gen)
    enc = VideoEncoder(sd, bench.model_cfg(H, K, T), dtype=torch.float16, tower_dtype=torch.bfloat16, device=dev,
                       tower_batch=512, tower_res_dtype=torch.float16)
    del sd
    enc.cfg["audio_input"] = True
    enc.c.audio_proj = Wt.make_lin(torch.randn(H, 768, device=dev, generator=gen) * 0.02, torch.zeros(H, device=dev),
                                   torch.float16, dev)
    enc.beats = BeatsEncoder(random_beats_state(BEATS_ITER3_CFG), BEATS_ITER3_CFG, dtype=torch.float16, device=dev)
    torch.cuda.empty_cache()
    wav = (0.1 * torch.randn(1, 16000 * T, device=dev, generator=gen)).half()
    vs = bench.synth_video(0, T, 384, dev, torch.bfloat16)
    vd = bench.synth_video(0, T, 378, dev, torch.bfloat16, seed=4321)
    prompt = [101] + list(range(2000, 2010)) + [102]

    def go(keep=None):
        return enc.encode_video(vs, vd, (384, 384), budget_text_len=64, n_text_tokens=64, prompt_ids=prompt, frame_cap=T,
                                audio={"audio_wav": wav}, keep=keep)
    keep = {}
    a = go(keep)
    b = go()
    assert torch.equal(a, b)                                            # run to run
    enc.tower_batch = 96                                                # 5 batches of 96 + a ragged tail of 32
    c = go()
    assert torch.equal(a, c)                                            # tower-batch invariance
    plan = keep["plan"]
    n_static, n_comp = len(plan["chunks"]), len(plan["comp_frames"])
    assert len(keep["selected"]) == T and n_static + n_comp == T and n_comp > 0
    assert a.shape[0] == n_static * (N + 50 + 1) + n_comp * (K + 1) and torch.isfinite(a.float()).all()
    rows = a.float()
    comp = torch.from_numpy((plan.kind == 1).nonzero()[0]).to(dev)
    assert comp.numel() == n_comp * K
    assert (rows[comp].norm(dim=1) - 1).abs().max().item() < 2e-3       # unit-norm compressed context rows (fp16 storage)
    seps = torch.from_numpy((plan.kind == 2).nonzero()[0]).to(dev)
    assert seps.numel() == T
    assert torch.equal(a[seps], enc.c.frame_seg[0, :H].to(a.dtype).expand(seps.numel(), H))


# (bf16 operands over the fp32 stream - 2.4e-2 measured against a 3.7e-2 bound in rounds 1-5 - left the suite in round 6: the fold is
#  a non-default alternative and the suite's time budget goes to the configurations a caller can reach by default)
@pytest.mark.parametrize("dtype,tol,stream16", [(torch.float16, 4.5e-3, False),     # measured 2.9e-3 (DINOv2)
                                                (torch.float16, 6e-3, True)])       # measured 4.8e-3
def test_fullsize_ln_fusion_matches_layernorm_kernel(full_sd, dtype, tol, stream16, monkeypatch):
    """VideoEncoder(ln_fuse=True) (pre-LayerNorms folded into the neighbouring GEMMs: 16-bit row copy + per-slot statistics out of the
    residual-stream GEMM, (mean, rstd) folded into the next GEMM's epilogue) against the LayerNorm-kernel path, both
    towers at full depth / width; and the fused path is batch invariant bit for bit like everything else.
    stream16 (round 6): the same over the fp16 residual stream - the consumers read the stream itself, the read-modify-write
    epilogues emit only the partials, no LayerNorm kernel inside the layer loop - and tdc_vit_fwd == the per-kernel sequence."""
    import bench
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    sd, cfg, _, _ = full_sd
    vs = bench.synth_video(0, 3, 384, "cuda:0", dtype)
    vd = bench.synth_video(0, 3, 378, "cuda:0", dtype, seed=4321)
    outs = {}
    for fuse in ("0", "1"):
        enc = VideoEncoder(sd, cfg, dtype=dtype, device="cuda:0", tower_batch=3, ln_fuse=fuse == "1",
                           tower_res_dtype=dtype if stream16 else None)
        assert all(bool(t.fused) == (fuse == "1") for t in enc.towers.values())
        outs[fuse] = (enc.tower("siglip", vs).float(), enc.tower("dino", vd).float())
        if fuse == "1":
            enc.tower_batch = 2
            assert torch.equal(enc.tower("dino", vd).float(), outs[fuse][1])
            if stream16:
                enc.tower_batch, enc.native_towers = 3, False
                assert torch.equal(enc.tower("siglip", vs).float(), outs[fuse][0])
                assert torch.equal(enc.tower("dino", vd).float(), outs[fuse][1])
        del enc
        torch.cuda.empty_cache()
    for a, b in zip(outs["0"], outs["1"]):
        d = ((a - b).abs().max() / a.abs().max()).item()
        print("ln_fuse vs LayerNorm kernel %s: %.3e of max|ref|" % (dtype, d))
        assert d < tol


@pytest.mark.parametrize("level", [1, 2, 3])
def test_fullsize_fp8_towers_close_to_bf16(level):
    """BASELINE config 5's fp8 MFMA path at full depth / width against the bf16 path on the same random-init weights.
    Level 1 (e4m3 operands for the LayerNorm-fed qkv / fc1 GEMMs): the emitted tokens differ by ~1 % RMS (measured
    1.2e-2) and the segment selection - the integer part of the path - is unchanged; level 2 (out-proj / fc2 too, inputs
    quantised per row by tdc_quantize_rows_fp8): the towers move by 8.8 % / 24 % RMS (level 1: 5.8 % / 19 %) - e4m3 has
    3 mantissa bits and random-init weights offer nothing to average the noise against - which is enough to reorder the
    near-tied adjacent-frame similarities of the synthetic video, so only the towers are compared there."""
    import bench
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    dev = torch.device("cuda", 0)
    H, K, T = 3584, 144, 40
    cfg = bench.model_cfg(H, K, T)
    sd = bench.random_state_dict(H, K, dev, torch.Generator(device=dev).manual_seed(0))
    vs = bench.synth_video(0, T, 384, dev, torch.bfloat16, scene_len=5)
    vd = bench.synth_video(0, T, 378, dev, torch.bfloat16, seed=4321, scene_len=5)
    outs = {}
    for fp8 in (0, level):
        enc = VideoEncoder(sd, cfg, dtype=torch.bfloat16, device=dev, tower_batch=20, fp8_towers=fp8)
        assert all(int(t.fp8) == fp8 for t in enc.towers.values())
        keep = {}
        vis = run(enc, vs, vd, keep)
        outs[fp8] = (vis.float(), keep["seg_indices"], keep["dino_feat"].float(), keep["siglip_feat"].float())
        del enc
        torch.cuda.empty_cache()

    def rms(i):
        a, b = outs[0][i], outs[level][i]
        return ((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()).item()
    print("fp8 level %d vs bf16 at full size: DINOv2 tower rel RMS %.3e, SigLIP %.3e" % (level, rms(2), rms(3)))
    assert rms(2) < (0.3 if level == 1 else 0.4) and rms(3) < (0.1 if level == 1 else 0.15)
    if level == 1:
        assert outs[0][1] == outs[1][1] and outs[0][0].shape == outs[1][0].shape
        print("   emitted tokens rel RMS %.3e" % rms(0))
        assert rms(0) < 4e-2
