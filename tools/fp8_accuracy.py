"""How far is the fp8-operand tower path (e4m3 qkv / fc1 inputs and weights) from the 16-bit path at FULL depth and width
(random-init SigLIP-so400m / DINOv2-giant, bench weights)?  Prints the RMS and max difference relative to the bf16 output,
per tower and at the end of the pipeline (compressed context tokens)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd.pipeline import VideoEncoder  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    H, K, T = 3584, 144, 40
    cfg = bench.model_cfg(H, K, T)
    gen = torch.Generator(device=dev).manual_seed(0)
    sd = bench.random_state_dict(H, K, dev, gen)
    vs = bench.synth_video(0, T, 384, dev, torch.bfloat16, scene_len=5)
    vd = bench.synth_video(0, T, 378, dev, torch.bfloat16, seed=4321, scene_len=5)
    outs = {}
    for fp8 in (0, 1, 2):
        enc = VideoEncoder(sd, cfg, dtype=torch.bfloat16, device=dev, tower_batch=20, fp8_towers=fp8)
        keep = {}
        vis = enc.encode_video(vs, vd, (384, 384), budget_text_len=64, n_text_tokens=64,
                               prompt_ids=[101] + list(range(2000, 2010)) + [102], keep=keep)
        outs[fp8] = (keep["siglip_feat"].float(), keep["dino_feat"].float(), vis.float(), keep["seg_indices"])
        del enc
        torch.cuda.empty_cache()
    for level in (1, 2):
        for name, i in (("siglip tower", 0), ("dino tower", 1), ("emitted tokens", 2)):
            a, b = outs[0][i], outs[level][i]
            if a.shape != b.shape:
                print("%-15s shapes differ %s vs %s" % (name, tuple(a.shape), tuple(b.shape)))
                continue
            rms = ((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()).item()
            print("level %d %-15s fp8 vs bf16: rel RMS diff %.3e, max diff / max|bf16| %.3e"
                  % (level, name, rms, ((a - b).abs().max() / a.abs().max()).item()))
        print("level %d segment selection identical:" % level, outs[0][3] == outs[level][3])


if __name__ == "__main__":
    main()
