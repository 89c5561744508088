"""CPU, world_size 2 (gloo): the frame-sharded path of tdc-video_amd/dist.py (similarity all-gather, key-frame query
hand-off across the rank boundary, emitted-token all-gather) must reproduce the serial orchestration bit for bit.
The numerical engine is replaced by a deterministic CPU test double: the product engine needs the HIP library."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import tdc_video_amd  # noqa: F401
from tdc_video_amd import pipeline, segment as seg
from tdc_video_amd.dist import ShardedVideoEncoder, split_plan


class FakeEngine:
    """Per-frame deterministic arithmetic standing in for the HIP engine (same interface as VideoEncoder)."""
    dtype = torch.float32

    def __init__(self, K=3, H=8, P=4, N=5, max_len=10 ** 9, **cfg):
        self.K, self.H, self.P, self.N = K, H, P, N
        self.cfg = dict(tokenizer_model_max_length=max_len, context_token_num=K, max_num_segments=24,
                        hidden_size=H)
        self.cfg.update(cfg)

    def tower(self, name, px):  # px [B, 3, 2, 2] -> [B*P, H]
        B = px.shape[0]
        base = px.reshape(B, -1)[:, : self.P].reshape(B * self.P, 1)
        return base * torch.arange(1, self.H + 1).float()[None] * (1.0 if name == "dino" else 0.5)

    def sims_tensor(self, dino, T):
        f = dino.reshape(T, -1)
        return torch.nn.functional.cosine_similarity(f[:-1], f[1:], dim=1)

    def connector(self, sig, dino, T, sizes, keep=None):
        x = (sig.reshape(T, self.P, self.H).sum(1) + dino.reshape(T, self.P, self.H).mean(1))      # [T, H]
        X = x[:, None, :] + torch.arange(self.N).float()[None, :, None]
        return X.reshape(T * self.N, self.H), [(1, self.N - 1)] * T

    def local_audio(self, audio, sample_indices, T, lo=0, hi=None):
        """audio = {"per_second": [S, Na, H]}: the token block of kept frame i is the mean over the seconds that a1 folds
        into it (its own second and the dropped ones that follow) - a stand-in for a20 that, like a20, depends on
        `sample_indices` and is local to the frame."""
        if audio is None:
            return None
        sec = audio["per_second"]
        kept = [i for i, v in enumerate(sample_indices) if v == 1]
        out = []
        for j in range(lo, T if hi is None else hi):
            a = kept[j]
            b = kept[j + 1] if j + 1 < len(kept) else len(sample_indices)
            out.append(sec[a:b].mean(0))
        return torch.stack(out) if out else sec[0:0]

    def with_audio(self, X, T, N, audio):
        if audio is None:
            return X, N
        Na = audio.shape[1]
        Xf = torch.cat([X.reshape(T, N, self.H), audio * 0.25], 1)
        return Xf.reshape(T * (N + Na), self.H), N + Na

    def make_queries(self, Xf, N, Nf, key_rows):
        rows = [Xf[r * Nf:(r * Nf + N)].mean(0, keepdim=True) + torch.arange(self.K).float()[:, None] for r in key_rows]
        return torch.cat(rows, 0)

    def learned_queries(self):
        return torch.arange(self.K * self.H).float().reshape(self.K, self.H) * 0.01

    def query_width(self):
        return self.H

    def compress_frames(self, Xf, Nf, frame_rows, qtable, qsrc, prompt_ids, keep=None):
        out = []
        for f, q in zip(frame_rows, qsrc):
            enc = Xf[f * Nf:(f + 1) * Nf]
            out.append(qtable[q * self.K:(q + 1) * self.K] * 2.0 - enc.mean(0, keepdim=True)
                       + (0.0 if prompt_ids is None else 0.125 * len(prompt_ids)))
        return torch.cat(out, 0)

    def emit(self, Xf, comp, pairs, splice=None):
        sep = torch.full((1, self.H), -7.0)
        tabs = [Xf, comp if comp is not None else sep, sep]
        return torch.stack([tabs[int(k)][int(r)] for k, r in pairs]) if len(pairs) else torch.zeros(0, self.H)


class NoisyEngine(FakeEngine):
    """FakeEngine whose ordinary DINO tower is a slightly perturbed copy of a `precise` one (as bf16 operands are of fp16 ones) and
    which offers the precise tower for the a5 selection refinement (segment.selection_band / select_refined): the per-frame
    perturbation moves the adjacent-frame similarities by less than selection_eps."""
    selection_eps = 2e-3
    selection_max_fraction = 1.0         # these cases refine wide bands on purpose (the product's default caps them at 1/8)

    def precise_dino(self, px):
        return FakeEngine.tower(self, "dino", px)

    def tower(self, name, px):
        f = FakeEngine.tower(self, name, px)
        if name == "dino":
            f = f * (1.0 + 2e-4 * torch.sin(977.0 * f))
        return f

    def pair_sims(self, feats, pairs):
        a = torch.stack([feats[i].reshape(-1) for i, _ in pairs])
        b = torch.stack([feats[j].reshape(-1) for _, j in pairs])
        return torch.nn.functional.cosine_similarity(a, b, dim=1)


def make_video(T):
    g = torch.Generator().manual_seed(5)
    base = torch.rand(3, 2, 2, generator=g)
    fr = []
    for t in range(T):
        if t % 5 == 0:
            base = torch.rand(3, 2, 2, generator=g) + t
        fr.append(base + 0.01 * torch.rand(3, 2, 2, generator=g))
    return torch.stack(fr)


CASES = {
    # name: (T0, tokenizer_model_max_length, N, cfg overrides, audio?, frame_cap)
    "plain61": (61, 10 ** 9, 5, {}, False, 10 ** 6),
    "plain40": (40, 10 ** 9, 5, {}, False, 10 ** 6),
    "passthrough20": (20, 10 ** 9, 5, {}, False, 10 ** 6),               # T <= 25: every frame static, no Q-Former
    "clip61": (61, 1250, 50, {}, False, 10 ** 6),                         # a19 tail clipping active, a1 cap not
    "learned61": (61, 10 ** 9, 5, {"query_type": "learned"}, False, 10 ** 6),
    "nostatic40": (40, 10 ** 9, 5, {"add_static": False}, False, 10 ** 6),
    "notext61": (61, 10 ** 9, 5, {"text_input": False}, False, 10 ** 6),
    "audio61": (61, 10 ** 9, 5, {"audio_input": True}, True, 10 ** 6),
    "cap90": (90, 10 ** 9, 5, {}, False, 37),                             # a1 cap: 90 frames -> 37 kept
    "cap90_audio": (90, 10 ** 9, 5, {"audio_input": True}, True, 37),     # ... dropped seconds folded into kept frames
    "budget120": (120, 20 + 4 + 16 + 3 * 30, 5, {}, False, 10 ** 6),      # a1 budget (get_max_num_frames) + a19 clip
    # a5 refinement: near-tied similarities under a perturbed tower, the band's pairs re-encoded by the precise one - incl. pairs
    # that cross a rank boundary (precise boundary features exchanged) and ranks that own no band pair
    "refine61": (61, 10 ** 9, 5, {"refine": True}, False, 10 ** 6),
    "refine90_few": (90, 10 ** 9, 5, {"refine": True, "max_num_segments": 17}, False, 10 ** 6),
}


def _case(name):
    T0, max_len, N, over, with_audio, cap = CASES[name]
    over = dict(over)
    eng = (NoisyEngine if over.pop("refine", False) else FakeEngine)(max_len=max_len, N=N, **over)
    vid = make_video(T0)
    audio = None
    if with_audio:
        g = torch.Generator().manual_seed(11)
        audio = {"per_second": torch.rand(T0, 2, eng.H, generator=g)}
    return eng, vid, audio, cap


def _run_rank(eng, vid, audio, cap, rank, world, comm=None, halo=False):
    sh = ShardedVideoEncoder(eng, rank, world, comm=comm)
    fp = sh.frame_plan(vid.shape[0], budget_text_len=4, frame_cap=cap, halo=halo)
    return sh.encode_video(vid[fp["siglip_frames"]], vid[fp["dino_frames"]], fp["T"], (384, 384), n_text_tokens=4,
                           prompt_ids=[1, 2], audio=audio, sample_indices=fp["sample_indices"],
                           recompute_halo=fp["recompute_halo"])


def _serial(name):
    eng, vid, audio, cap = _case(name)
    return pipeline.encode_video_with(eng, vid, vid, (384, 384), budget_text_len=4, n_text_tokens=4, prompt_ids=[1, 2],
                                      audio=audio, frame_cap=cap)


def _worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng, vid, audio, cap = _case(name)
        out = _run_rank(eng, vid, audio, cap, rank, world)
        q.put((rank, out.numpy()))  # by value: the worker exits before the parent reads
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("name", ["plain61", "passthrough20", "clip61", "learned61", "cap90_audio", "budget120", "refine61"])
def test_sharded_equals_serial_world2(name):
    """two processes over gloo: the product transport (TorchComm) end to end"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: torch.from_numpy(a) for r, a in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _serial(name)
    if name == "clip61":
        eng, vid, _, _ = _case(name)
        full = pipeline.compress_with(FakeEngine(N=eng.N), *eng.connector(eng.tower("siglip", vid), eng.tower("dino", vid),
                                                                           61, None)[:1], 61, eng.N,
                                      seg.select_segments(eng.sims_tensor(eng.tower("dino", vid), 61).tolist(), 24),
                                      [1, 2], 10 ** 9)
        assert want.shape[0] <= 1250 - 16 - 4 < full.shape[0]     # the a19 clip really happened
    for r in range(world):
        assert res[r].shape == want.shape
        assert torch.equal(res[r], want), "rank %d differs" % r


def _threads(name, world, halo=False):
    import threading
    from util import ThreadComm
    hub = ThreadComm.Hub(world)
    out, err = [None] * world, []

    def run(r):
        try:
            eng, vid, audio, cap = _case(name)
            out[r] = _run_rank(eng, vid, audio, cap, r, world, comm=ThreadComm(hub, r), halo=halo)
        except BaseException as ex:      # noqa: BLE001 - release the peers, report in the main thread
            err.append((r, ex))
            hub.bar.abort()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    real = [e for e in err if not isinstance(e[1], threading.BrokenBarrierError)]
    assert not err, real or err
    return out


@pytest.mark.parametrize("world", [3, 4, 8])
@pytest.mark.parametrize("name", sorted(CASES))
def test_sharded_equals_serial_any_world(name, world):
    """every configuration x world size through the in-process transport (same ShardedVideoEncoder code, threads as
    ranks): 8 ranks incl. ranks that own no key frame, chunks that straddle two boundaries, ranks without compressed frames"""
    want = _serial(name)
    for r, got in enumerate(_threads(name, world)):
        assert got.shape == want.shape and torch.equal(got, want), "rank %d of %d differs" % (r, world)


@pytest.mark.parametrize("name", ["plain61", "cap90_audio", "refine61"])
def test_boundary_frame_reencoded_instead_of_exchanged(name):
    """frame_plan(halo=True): the caller hands every rank the next rank's first frame and it is re-encoded locally (round 1's
    form) - same stream as the exchange of its DINOv2 features"""
    want = _serial(name)
    for r, got in enumerate(_threads(name, 4, halo=True)):
        assert got.shape == want.shape and torch.equal(got, want), "rank %d differs" % r


def test_selection_refinement_gives_the_precise_towers_selection():
    """the refine cases: the perturbed tower's similarities leave a non-empty band, and the refined selection is what ranking the
    PRECISE tower's similarities selects - through the serial orchestration (info["seg_indices"])"""
    for name in ("refine61", "refine90_few"):
        eng, vid, audio, cap = _case(name)
        mns = eng.cfg["max_num_segments"]
        T = vid.shape[0]
        noisy = eng.sims_tensor(eng.tower("dino", vid), T).tolist()
        precise = eng.sims_tensor(eng.precise_dino(vid), T).tolist()
        assert max(abs(a - b) for a, b in zip(noisy, precise)) < eng.selection_eps
        band = seg.selection_band(noisy, mns, eng.selection_eps)
        assert len(band) >= 2 and len(band) < T - 1
        info = {}
        pipeline.encode_video_with(eng, vid, vid, (384, 384), budget_text_len=4, n_text_tokens=4, prompt_ids=[1, 2], frame_cap=cap,
                                   info=info)
        assert info["refined_pairs"] == band
        assert info["seg_indices"] == seg.select_segments(precise, mns)
    # a band that covers a rank boundary exists in the sharded runs of these cases (world 4 / 8 over 61 frames)
    eng, vid, _, _ = _case("refine61")
    band = set(seg.selection_band(eng.sims_tensor(eng.tower("dino", vid), 61).tolist(), 24, eng.selection_eps))
    assert any((h - 1) in band for (l, h) in seg.shard_ranges(61, 4)[:-1])


def test_selection_band_property():
    """selection_band / select_refined: for similarities within eps of the precise ones, re-ranking only the band by the precise
    values selects exactly what ranking all precise values selects (ties and dense clusters included); a gap wider than 2 eps at
    the decisive rank needs nothing"""
    import random
    rng = random.Random(1)
    for trial in range(1500):
        n, mns = rng.choice([30, 60, 511]), rng.choice([2, 5, 24])
        eps, spread = rng.choice([1e-3, 3e-3, 1e-2]), rng.choice([0.02, 0.25, 1.0])
        ref = [rng.random() * spread for _ in range(n)]
        if trial % 7 == 0:
            for k in range(0, n - 1, 3):
                ref[k + 1] = ref[k]
        noisy = [v + (rng.random() * 2 - 1) * eps * 0.999 for v in ref]
        band = seg.selection_band(noisy, mns, eps)
        assert seg.select_refined(noisy, mns, eps, band, [ref[i] for i in band]) == seg.select_segments(ref, mns)
    assert seg.selection_band([0.1, 0.2, 0.5, 0.9], 2, 0.01) == [] and seg.selection_band([0.1, 0.2], 2, 0.01) == []
    assert seg.selection_band([0.1, 0.2, 0.21, 0.9], 2, 0.01) == [1, 2] and seg.band_frames([1, 2, 7]) == [1, 2, 3, 7, 8]
    # the cost cap: a band is refined when its frames number at most max(8, fraction x T)
    assert seg.band_allowed([1, 2, 7], 512, 0.125) and seg.band_allowed(list(range(0, 63)), 512, 0.125)
    assert not seg.band_allowed(list(range(0, 64)), 512, 0.125) and not seg.band_allowed([], 512, 0.125)
    assert seg.band_allowed([3, 4, 5, 6, 7, 8, 9], 40, 0.125) and not seg.band_allowed(list(range(3, 11)), 40, 0.125)


def test_selection_refinement_is_capped_on_a_plateau():
    """a clip with fewer scene changes than max_num_segments: the decisive rank lies in a plateau of near-identical similarities, the
    band covers most of the video - beyond the cap the fast tower's ranking stands (serial and sharded alike)"""
    eng, vid, audio, cap = _case("refine61")
    eng.selection_max_fraction = 0.125
    info = {}
    want = pipeline.encode_video_with(eng, vid, vid, (384, 384), budget_text_len=4, n_text_tokens=4, prompt_ids=[1, 2], frame_cap=cap,
                                      info=info)
    noisy = eng.sims_tensor(eng.tower("dino", vid), 61).tolist()
    assert info["refined_pairs"] == [] and info["refine_skipped_pairs"] > 8 and info["seg_indices"] == seg.select_segments(noisy, 24)
    import threading
    from util import ThreadComm
    hub = ThreadComm.Hub(4)
    out = [None] * 4

    def run(r):
        e2, v2, _, c2 = _case("refine61")
        e2.selection_max_fraction = 0.125
        out[r] = _run_rank(e2, v2, None, c2, r, 4, comm=ThreadComm(hub, r))
    ts = [threading.Thread(target=run, args=(r,)) for r in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert all(o is not None and torch.equal(o, want) for o in out)


def test_cases_exercise_what_they_claim():
    eng, vid, audio, cap = _case("cap90_audio")
    sh = ShardedVideoEncoder(eng, 1, 4, comm=object())
    fp = sh.frame_plan(90, 4, cap)
    assert fp["T"] == 37 and sum(fp["sample_indices"]) == 37 and len(fp["sample_indices"]) == 90
    assert fp["dino_frames"] == fp["siglip_frames"]
    fp = sh.frame_plan(90, 4, cap, halo=True)
    assert fp["dino_frames"][:-1] == fp["siglip_frames"] and len(fp["dino_frames"]) == len(fp["siglip_frames"]) + 1
    eng, vid, _, cap = _case("budget120")
    assert seg.get_max_num_frames(4, eng.cfg) < 120                 # the token budget, not frame_cap, limits the frames
    # with a video_index (decoded frames sit on a subset of the seconds) the kept seconds follow it
    vi = [1 if i % 2 == 0 else 0 for i in range(180)]
    eng = _case("cap90")[0]
    fp = ShardedVideoEncoder(eng, 0, 2, comm=object()).frame_plan(90, 4, 37, video_index=vi)
    assert len(fp["sample_indices"]) == 180 and sum(fp["sample_indices"]) == 37
    assert all(vi[i] == 1 for i, v in enumerate(fp["sample_indices"]) if v == 1)


def test_split_plan_partitions_the_stream():
    T, N, K = 61, 5, 3
    segi = seg.select_segments([((i * 37) % 60) / 60.0 for i in range(T - 1)], 24)
    plan = seg.emit_plan(T, N, K, segi, 10 ** 9)
    for world in (1, 2, 4, 8):
        ranges = seg.shard_ranges(T, world)
        per, comp_local = split_plan(plan, ranges, N, K)
        assert sum(len(p) for p in per) == len(plan["src"])
        # rows stay inside each rank's local tables
        for r, pairs in enumerate(per):
            lo, hi = ranges[r]
            assert all(row < (hi - lo) * N for k, row in pairs if k == 0)
            assert all(row < len(comp_local[r]) * K for k, row in pairs if k == 1)


# ------------------------------------------------------------------------------------------------ through the mixin
class MixinEngine(FakeEngine):
    """FakeEngine with what the boundary (model.CambrianMetaForCausalLM) touches besides the stage methods: a device, the SVA
    grid side (2 x 2 -> 2 * (2 + 1) = 6 rows per frame, consistent with `connector`'s sizes) and `encode_video`."""
    dev = torch.device("cpu")
    side = 2

    def __init__(self, **kw):
        super().__init__(N=6, **kw)

    def connector(self, sig, dino, T, sizes, keep=None):
        X, _ = super().connector(sig, dino, T, sizes, keep)
        return X, [(2, 2)] * T

    def encode_video(self, px_siglip, px_dino, image_size, budget_text_len, n_text_tokens, prompt_ids, audio=None,
                     frame_cap=224, keep=None, splice=None, video_index=None, info=None):
        return pipeline.encode_video_with(self, px_siglip, px_dino, image_size, budget_text_len, n_text_tokens, prompt_ids,
                                          audio, frame_cap, keep, splice, video_index, info)


def _mixin_call(shard, T0=90, cap=37):
    """prepare_inputs_labels_for_multimodal on a stub LM whose engine is the CPU double; config.tdc_frame_cap lifts /
    lowers both frame caps, config.tdc_shard_frames routes through dist.ShardedVideoEncoder when a process group exists."""
    from test_host_logic import build_stub_lm, tiny_config
    torch.manual_seed(3)
    lm = build_stub_lm(tiny_config(hidden_size=8, context_token_num=3, tdc_frame_cap=cap, tdc_shard_frames=shard,
                                   tokenizer_model_max_length=10 ** 9))
    lm.get_model()._tdc_encoder = MixinEngine(K=3, H=8)
    vid = make_video(T0)
    ids = torch.tensor([[11, 12, -200, 13, 14, 15]])
    with torch.inference_mode():            # as the reference's generate() calls it (cambrian_qwen.py:401)
        out = lm.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [vid[None], vid[None]],
                                                      image_sizes=[(384, 384)], video_indices=[None], prompts=[[1, 2]],
                                                      audios=[None])
    return out


def _mixin_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = _mixin_call(True)
        q.put((rank, out[4].numpy(), out[8]))
    finally:
        dist.destroy_process_group()


def test_mixin_shards_frames_world2_equals_serial():
    """config.tdc_shard_frames: two gloo processes call the boundary with the same 90-frame video; each encodes its half of
    the 37 frames config.tdc_frame_cap keeps and both return the serial call's inputs_embeds / final_size, bit for bit"""
    want = _mixin_call(False)
    assert want[4].shape[1] > 6 + 37 and len(want[8]) == 37          # the cap of the config key, not the constant 224
    assert len(_mixin_call(False, cap=224)[8]) == 90
    # without a process group the key alone changes nothing
    assert torch.equal(_mixin_call(True)[4], want[4])
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mixin_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (torch.from_numpy(a), fs) for r, a, fs in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert torch.equal(res[r][0], want[4]), "rank %d differs" % r
        assert [tuple(x) for x in res[r][1]] == [tuple(x) for x in want[8]]
