"""Shared helpers for the tests: fixture loading and the oracle import (tests are allowed to use oracle/)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tdc_oracle as oracle  # noqa: E402


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    W, other = {}, {}
    for k in z.files:
        if k.startswith("w::"):
            kk = k[3:]
            if kk.startswith("model."):
                kk = kk[6:]
            W[kk] = torch.from_numpy(z[k])
        else:
            other[k] = z[k]
    return W, other


def embed_fn(other):
    ids = [int(i) for i in other["used_embed_ids"]]
    rows = torch.from_numpy(other["used_embed_rows"])
    table = {i: rows[n] for n, i in enumerate(ids)}

    def fn(x):
        if len(x) == 0:
            return rows[0:0]
        return torch.stack([table[int(i)] for i in x])
    return fn


def pipeline_cfg(other):
    cfg = json.loads(str(other["cfg_json"]))
    cfg.update(dino_heads=4, siglip_heads=4, qformer_heads=4)
    return cfg


class ThreadComm:
    """In-process transport with the interface of tdc_video_amd.dist.TorchComm: `world` threads of ONE process exchange
    tensors through shared slots and barriers.  Test infrastructure only: it lets a one-GPU box (which admits at most six
    GPU processes) rehearse the data flow of an 8-rank job; the product transport is torch.distributed."""

    class Hub:
        def __init__(self, world):
            import threading
            self.world = world
            self.bar = threading.Barrier(world)
            self.slots = [None] * world
            self.mail = {}
            self.lock = threading.Lock()

    def __init__(self, hub, rank):
        self.hub, self.rank, self.world = hub, rank, hub.world

    def all_gather(self, t):
        h = self.hub
        h.slots[self.rank] = t.clone()
        h.bar.wait()
        out = [s.clone() for s in h.slots]
        h.bar.wait()
        return out

    def exchange(self, sends, recvs):
        h = self.hub
        with h.lock:
            for t, dst in sends:
                h.mail.setdefault((self.rank, dst), []).append(t.clone())
        h.bar.wait()
        for buf, src in recvs:
            with h.lock:
                t = h.mail[(src, self.rank)].pop(0)
            buf.copy_(t)
        h.bar.wait()
