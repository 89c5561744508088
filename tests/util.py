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
