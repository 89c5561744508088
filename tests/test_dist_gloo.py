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

    def __init__(self, K=3, H=8, P=4, N=5, max_len=10 ** 9):
        self.K, self.H, self.P, self.N = K, H, P, N
        self.cfg = dict(tokenizer_model_max_length=max_len, context_token_num=K, max_num_segments=24,
                        hidden_size=H)

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

    def with_audio(self, X, T, N, audio):
        return X, N

    def make_queries(self, Xf, N, Nf, key_rows):
        rows = [Xf[r * Nf:(r * Nf + N)].mean(0, keepdim=True) + torch.arange(self.K).float()[:, None] for r in key_rows]
        return torch.cat(rows, 0)

    def query_width(self):
        return self.H

    def compress_frames(self, Xf, Nf, frame_rows, qtable, qsrc, prompt_ids, keep=None):
        out = []
        for f, q in zip(frame_rows, qsrc):
            enc = Xf[f * Nf:(f + 1) * Nf]
            out.append(qtable[q * self.K:(q + 1) * self.K] * 2.0 - enc.mean(0, keepdim=True))
        return torch.cat(out, 0)

    def emit(self, Xf, comp, pairs):
        sep = torch.full((1, self.H), -7.0)
        tabs = [Xf, comp if comp is not None else sep, sep]
        return torch.stack([tabs[k][r] for k, r in pairs]) if pairs else torch.zeros(0, self.H)


def make_video(T):
    g = torch.Generator().manual_seed(5)
    base = torch.rand(3, 2, 2, generator=g)
    fr = []
    for t in range(T):
        if t % 5 == 0:
            base = torch.rand(3, 2, 2, generator=g) + t
        fr.append(base + 0.01 * torch.rand(3, 2, 2, generator=g))
    return torch.stack(fr)


def _worker(rank, world, port, T, max_len, N, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = FakeEngine(max_len=max_len, N=N)
        vid = make_video(T)
        lo, hi = seg.shard_ranges(T, world)[rank]
        halo = 1 if rank < world - 1 else 0
        sh = ShardedVideoEncoder(eng, rank, world)
        out = sh.encode_video(vid[lo:hi], vid[lo:hi + halo], T, (384, 384), n_text_tokens=4, prompt_ids=[1, 2])
        q.put((rank, out.numpy()))  # by value: the worker exits before the parent reads
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("T,max_len,N", [(61, 10 ** 9, 5), (40, 10 ** 9, 5), (20, 10 ** 9, 5),
                                         (61, 1250, 50)])   # last: a19 tail clipping active, a1 cap not
def test_sharded_equals_serial_world2(T, max_len, N):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, max_len, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: torch.from_numpy(a) for r, a in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    eng = FakeEngine(max_len=max_len, N=N)
    vid = make_video(T)
    want = pipeline.encode_video_with(eng, vid, vid, (384, 384), budget_text_len=4, n_text_tokens=4, prompt_ids=[1, 2],
                                      frame_cap=10 ** 6)
    if max_len < 10 ** 9:
        full = pipeline.compress_with(FakeEngine(N=N), *eng.connector(eng.tower("siglip", vid), eng.tower("dino", vid),
                                                                       T, None)[:1], T, N,
                                      seg.select_segments(eng.sims_tensor(eng.tower("dino", vid), T).tolist(), 24),
                                      [1, 2], 10 ** 9)
        assert want.shape[0] <= max_len - 16 - 4 < full.shape[0]     # the a19 clip really happened
    for r in range(world):
        assert res[r].shape == want.shape
        assert torch.equal(res[r], want), "rank %d differs" % r


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
