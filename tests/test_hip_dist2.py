"""GPU: the frame-sharded HIP pipeline with world_size 2 - two processes sharing cuda:0, exchanges over gloo (RCCL needs
one GPU per rank; the collectives' payloads and order are identical) - must reproduce the serial encode bit for bit:
similarities all-gather -> same segmentation; key-frame query hand-off across the rank boundary; token all-gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import synth
from util import load_fixture, pipeline_cfg

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, name, q):
    import torch.distributed as dist
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    from tdc_video_amd.dist import ShardedVideoEncoder
    from tdc_video_amd import segment as seg
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        W, o = load_fixture(name)
        cfg = pipeline_cfg(o)
        enc = VideoEncoder(W, cfg, dtype=torch.float16, device="cuda:0", siglip_heads=4, dino_heads=4, qformer_heads=4)
        vid = torch.from_numpy(synth.video_from_basis(o["video_basis"], o["video_coef"])).cuda()
        T = vid.shape[0]
        ids = torch.from_numpy(o["input_ids"])[0]
        lo, hi = seg.shard_ranges(T, world)[rank]
        halo = 1 if rank < world - 1 else 0
        sh = ShardedVideoEncoder(enc, rank, world)
        out = sh.encode_video(vid[lo:hi].contiguous(), (vid + 0.01)[lo:hi + halo].contiguous(), T,
                              tuple(int(v) for v in o["image_size"]), len(ids) - 1, [int(i) for i in o["prompt_ids"]])
        serial = None
        if rank == 0:
            serial = enc.encode_video(vid, vid + 0.01, tuple(int(v) for v in o["image_size"]), len(ids), len(ids) - 1,
                                      [int(i) for i in o["prompt_ids"]]).float().cpu().numpy()
        q.put((rank, out.float().cpu().numpy(), serial))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("name", ["pipeline_T40.npz"])
def test_sharded_world2_on_one_gpu_equals_serial(name):
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
