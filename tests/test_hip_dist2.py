"""GPU: the frame-sharded HIP pipeline must reproduce the serial encode bit for bit (similarities all-gather -> same
segmentation; key-frame query hand-off across rank boundaries; local audio tokens; token all-gather) on the
reference-generated fixtures:
  * world 2 as two PROCESSES sharing cuda:0 with the product transport (dist.TorchComm) over gloo - RCCL needs one GPU
    per rank; the collectives' payloads and order are identical;
  * world 4 and 8 as threads of one process through the tests' in-process transport (a one-GPU box admits at most six
    GPU processes): the same ShardedVideoEncoder code, every rank with its own engine.
Fixtures: the plain 40-frame video, + audio (a20: N + 50 KV rows), query_type='learned', and the 260-frame video whose
frame budget caps it to 98 frames (a1 on every rank) and whose token budget clips the stream (a19)."""
import os
import socket
import threading

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import synth
from util import ThreadComm, load_fixture, pipeline_cfg

pytestmark = pytest.mark.gpu

NAMES = ["pipeline_T40.npz", "pipeline_T40_audio.npz", "pipeline_T40_learned.npz", "pipeline_T260.npz"]


def _setup(name):
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    W, o = load_fixture(name)
    cfg = pipeline_cfg(o)
    enc = VideoEncoder(W, cfg, dtype=torch.float16, device="cuda:0", siglip_heads=4, dino_heads=4, qformer_heads=4)
    vid = torch.from_numpy(synth.video_from_basis(o["video_basis"], o["video_coef"])).cuda()
    ids = torch.from_numpy(o["input_ids"])[0]
    audio = None
    if "audio_wav" in o:
        audio = {"beats_windows": synth.beats_windows(torch.from_numpy(o["audio_wav"].astype(np.float32)))}
    return enc, vid, ids, tuple(int(v) for v in o["image_size"]), [int(i) for i in o["prompt_ids"]], audio


def _sharded(enc, vid, ids, size, pid, audio, rank, world, comm=None):
    from tdc_video_amd.dist import ShardedVideoEncoder
    sh = ShardedVideoEncoder(enc, rank, world, comm=comm)
    fp = sh.frame_plan(vid.shape[0], budget_text_len=len(ids))
    return sh.encode_video(vid[fp["siglip_frames"]].contiguous(), (vid + 0.01)[fp["dino_frames"]].contiguous(), fp["T"],
                           size, len(ids) - 1, pid, audio=audio, sample_indices=fp["sample_indices"])


def _serial(enc, vid, ids, size, pid, audio):
    return enc.encode_video(vid, vid + 0.01, size, len(ids), len(ids) - 1, pid, audio=audio)


def _worker(rank, world, port, name, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        args = _setup(name)
        out = _sharded(*args, rank, world)
        serial = _serial(*args).float().cpu().numpy() if rank == 0 else None
        q.put((rank, out.float().cpu().numpy(), serial))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("name", NAMES)
def test_sharded_world2_processes_on_one_gpu_equals_serial(name):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    serial = None
    for _ in range(world):
        r, out, ser = q.get(timeout=300)
        res[r] = out
        if ser is not None:
            serial = ser
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert serial is not None
    for r in range(world):
        assert res[r].shape == serial.shape
        assert np.array_equal(res[r], serial), "rank %d differs from the serial encode" % r


def run_threads(world, make_rank_fn):
    """rank r runs make_rank_fn(r, comm) on its own thread; returns the per-rank results (raises the first real error)"""
    hub = ThreadComm.Hub(world)
    out, err = [None] * world, []

    def run(r):
        try:
            torch.cuda.set_device(0)
            out[r] = make_rank_fn(r, ThreadComm(hub, r))
        except BaseException as ex:      # noqa: BLE001 - release the peers, report in the main thread
            err.append((r, ex))
            hub.bar.abort()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(900)
    real = [e for e in err if not isinstance(e[1], threading.BrokenBarrierError)]
    if real:
        raise real[0][1]
    assert not err, err
    return out


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("name", NAMES)
def test_sharded_world4_8_threads_equals_serial(name, world):
    engines = [_setup(name) for _ in range(world)]            # one engine (weights, workspaces, caches) per rank
    want = _serial(*engines[0])
    got = run_threads(world, lambda r, comm: _sharded(*engines[r], r, world, comm=comm))
    for r in range(world):
        assert got[r].shape == want.shape and torch.equal(got[r], want), "rank %d of %d differs" % (r, world)
