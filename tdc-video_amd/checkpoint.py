"""Checkpoint loading for the path's parameters (reference: tdc/builder.py:168-172,243-257).

The released checkpoints (Hoar012/TDC-Qwen2-7B, TDC-Llama3_2-3B) are HF-style directories of safetensors / torch shards
whose keys are the reference's state-dict names (`model.mm_projector.0.weight`, `model.Qformer.bert...`, ...); the vision
towers come from their own HF repositories (`google/siglip-so400m-patch14-384`, `facebook/dinov2-giant`).  Only the keys
that belong to the video-encoding path are read; LLM weights are left to the caller's own loader."""
import glob
import json
import os

import torch

PATH_PREFIXES = ("mm_projector", "vision_sampler_", "vision_query", "image_newline", "frame_seg", "Qformer.bert.",
                 "vision_proj.", "query_proj.", "audio_proj.", "query_tokens")


def _iter_shards(ckpt_dir):
    idx = glob.glob(os.path.join(ckpt_dir, "*.index.json"))
    files = None
    if idx:
        wm = json.load(open(idx[0]))["weight_map"]
        files = sorted(set(wm.values()))
    else:
        files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(ckpt_dir, "*.safetensors")) +
                       glob.glob(os.path.join(ckpt_dir, "pytorch_model*.bin")) +
                       glob.glob(os.path.join(ckpt_dir, "non_lora_trainables.bin")))
    for f in files:
        path = os.path.join(ckpt_dir, f)
        if f.endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(path, framework="pt", device="cpu") as sf:
                for k in sf.keys():
                    yield k, (lambda sf=sf, k=k: sf.get_tensor(k))
        else:
            sd = torch.load(path, map_location="cpu", weights_only=True)
            for k, v in sd.items():
                yield k, (lambda v=v: v)


def _strip(k):
    for p in ("base_model.model.", "module."):
        if k.startswith(p):
            k = k[len(p):]
    return k[6:] if k.startswith("model.") else k


def read_path_state(ckpt_dir):
    """reference-named tensors of the path (without 'model.') found in `ckpt_dir`; everything else is skipped unread."""
    out = {}
    for k, get in _iter_shards(ckpt_dir):
        kk = _strip(k)
        if kk.startswith(PATH_PREFIXES):
            out[kk] = get()
    return out


def read_tower_state(tower_dir):
    """HF SiglipVisionModel / Dinov2Model weights (4.46 'vision_model.' prefix and 5.x names both accepted); the SigLIP
    pooling head / post_layernorm and the DINOv2 mask token are not on the path."""
    out = {}
    for k, get in _iter_shards(tower_dir):
        kk = k.replace("vision_model.", "")
        if kk.startswith(("head.", "post_layernorm", "text_model", "logit_")) or "mask_token" in kk:
            continue
        out[kk] = get()
    return out


def load_path_weights(meta_model, ckpt_dir, siglip_dir=None, dino_dir=None, strict=True):
    """Fill a `model.CambrianMetaModel` (connector, Q-Former, projections) from a released checkpoint directory and the
    towers from their HF directories; returns (missing, unexpected) key lists of the connector part."""
    sd = read_path_state(ckpt_dir)
    res = meta_model.load_state_dict(sd, strict=False)
    missing = [k for k in res.missing_keys if k.startswith(PATH_PREFIXES)]
    if strict and missing:
        raise KeyError("checkpoint %s lacks %d path tensors, e.g. %s" % (ckpt_dir, len(missing), missing[:3]))
    towers = meta_model.get_vision_tower_aux_list()
    for t, d in zip(towers, (siglip_dir, dino_dir)):
        if d is not None:
            t.load_model(state_dict=read_tower_state(d))
    meta_model._tdc_encoder = None        # weights changed: rebuild the device engine on next use
    return missing, list(res.unexpected_keys)
