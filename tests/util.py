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


def write_released_style_checkpoint(W, root):
    """Lay the fixture weights out the way the released checkpoints are (tdc/builder.py:168-172,243-257): a
    directory of sharded safetensors + index whose keys carry the reference's 'model.' prefix (with unrelated LLM tensors
    mixed in) and one HF directory per tower (transformers-4.46 'vision_model.' prefix for SigLIP, bare names for DINOv2).
    Returns (ckpt_dir, siglip_dir, dino_dir)."""
    from safetensors.torch import save_file
    root = str(root)
    path_sd = {"model." + k: v.contiguous() for k, v in W.items() if not k.startswith("vision_tower_aux_list")}
    keys = sorted(path_sd)
    half = len(keys) // 2
    d = os.path.join(root, "ckpt")
    os.makedirs(d)
    shard_a = {k: path_sd[k] for k in keys[:half]}
    shard_a["model.layers.0.self_attn.q_proj.weight"] = torch.zeros(4, 4)      # LLM tensor: must be skipped
    shard_b = {k: path_sd[k] for k in keys[half:]}
    shard_b["lm_head.weight"] = torch.zeros(4, 4)
    save_file(shard_a, os.path.join(d, "model-00001-of-00002.safetensors"))
    save_file(shard_b, os.path.join(d, "model-00002-of-00002.safetensors"))
    wm = {k: "model-00001-of-00002.safetensors" for k in shard_a}
    wm.update({k: "model-00002-of-00002.safetensors" for k in shard_b})
    with open(os.path.join(d, "model.safetensors.index.json"), "w") as fh:
        json.dump({"weight_map": wm}, fh)
    tdirs = []
    for i, pre in enumerate(("vision_model.", "")):
        td = os.path.join(root, "tower%d" % i)
        os.makedirs(td)
        p = "vision_tower_aux_list.%d.vision_tower." % i
        sd = {pre + k[len(p):]: v.contiguous() for k, v in W.items() if k.startswith(p)}
        if i == 0:      # what the HF SigLIP checkpoint also holds and the path never reads
            sd["vision_model.head.probe"] = torch.zeros(1, 1, 4)
            sd["vision_model.post_layernorm.weight"] = torch.ones(4)
        save_file(sd, os.path.join(td, "model.safetensors"))
        tdirs.append(td)
    return d, tdirs[0], tdirs[1]


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

    def all_gather_into(self, out, t):
        parts = self.all_gather(t)
        n = t.shape[0]
        for r, part in enumerate(parts):
            out[r * n:(r + 1) * n].copy_(part)

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
