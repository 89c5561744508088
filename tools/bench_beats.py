"""BEATs encoder micro-benchmark: 10-second windows per second at the released-checkpoint dimensions
(12 x 768, 496 tokens / window), random weights.  python tools/bench_beats.py [--windows 52] [--bf16]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tdc_video_amd  # noqa: E402,F401
from tdc_video_amd import ops  # noqa: E402
from tdc_video_amd.beats import BEATS_ITER3_CFG, BeatsEncoder  # noqa: E402


def random_beats_state(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    C, E, F, H, nl = cfg["encoder_embed_dim"], cfg["embed_dim"], cfg["encoder_ffn_embed_dim"], cfg["encoder_attention_heads"], cfg["encoder_layers"]
    rn = lambda *s, std=0.02: torch.randn(*s, generator=g) * std
    W = {"patch_embedding.weight": rn(E, 1, 16, 16, std=0.06), "layer_norm.weight": torch.ones(E), "layer_norm.bias": torch.zeros(E),
         "post_extract_proj.weight": rn(C, E, std=0.04), "post_extract_proj.bias": rn(C),
         "encoder.pos_conv.0.weight_g": torch.ones(1, 1, cfg["conv_pos"]),
         "encoder.pos_conv.0.weight_v": rn(C, C // cfg["conv_pos_groups"], cfg["conv_pos"]), "encoder.pos_conv.0.bias": rn(C),
         "encoder.layer_norm.weight": torch.ones(C), "encoder.layer_norm.bias": torch.zeros(C)}
    emb = rn(cfg["num_buckets"], H, std=1.0)
    for i in range(nl):
        p = "encoder.layers.%d." % i
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            W[p + "self_attn." + n + ".weight"], W[p + "self_attn." + n + ".bias"] = rn(C, C, std=0.04), rn(C)
        W[p + "self_attn.grep_linear.weight"], W[p + "self_attn.grep_linear.bias"] = rn(8, C // H, std=0.2), rn(8)
        W[p + "self_attn.grep_a"] = torch.ones(1, H, 1, 1)
        W[p + "self_attn.relative_attention_bias.weight"] = emb
        W[p + "fc1.weight"], W[p + "fc1.bias"] = rn(F, C, std=0.04), rn(F)
        W[p + "fc2.weight"], W[p + "fc2.bias"] = rn(C, F, std=0.03), rn(C)
        for n in ("self_attn_layer_norm", "final_layer_norm"):
            W[p + n + ".weight"], W[p + n + ".bias"] = torch.ones(C), torch.zeros(C)
    return W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=52)      # a 512-second video
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dt = torch.bfloat16 if a.bf16 else torch.float16
    enc = BeatsEncoder(random_beats_state(BEATS_ITER3_CFG), BEATS_ITER3_CFG, dtype=dt, device="cuda:0")
    wav = (0.1 * torch.randn(1, 160000 * a.windows)).half().cuda()
    for _ in range(2):
        enc.window_features(wav)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        enc.window_features(wav)
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / a.iters
    # algorithmic GEMM + attention flops per window (496 tokens)
    L, C, F = 496, 768, 3072
    fl = 2 * L * (256 * 512 + 512 * 768 + 768 * 48 * 128) + 12 * (2 * L * (4 * C * C + 2 * C * F) + 4 * L * L * C)
    print("BEATs %s: %d windows (%.0f s of audio) in %.2f ms = %.0f windows/s, %.1f TFLOP/s algorithmic"
          % ("bf16" if a.bf16 else "fp16", a.windows, 10.0 * a.windows, dtm * 1e3, a.windows / dtm,
             fl * a.windows / dtm / 1e12))


if __name__ == "__main__":
    main()
