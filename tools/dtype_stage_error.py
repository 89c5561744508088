"""Where does the 16-bit error of the compressed tokens come from?  BASELINE config 1 at full depth (8 frames, 336 px,
max_num_segments = 2) against the fp32 oracle, with the towers and the rest of the path (connector + Q-Former) run in
independently chosen 16-bit types.  Lab tool (GPU box): python tools/dtype_stage_error.py > gpurun_out/dtype_stage_error.log
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
import tdc_oracle as orc  # noqa: E402
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd.pipeline import VideoEncoder, encode_video_with  # noqa: E402

PROMPT = [101] + list(range(2000, 2010)) + [102]


class Mixed:
    """engine for encode_video_with: towers of `a`, everything else of `b`"""

    def __init__(self, a, b):
        self.a, self.b = a, b
        self.cfg, self.K = b.cfg, b.K

    def tower(self, name, px):
        return self.a.tower(name, px).to(self.b.dtype)

    def __getattr__(self, n):
        return getattr(self.b, n)


def main():
    H, K, T, px, mns = 3584, 144, 8, 336, 2
    dev = "cuda:0"
    gen = torch.Generator(device=dev).manual_seed(0)
    sd = {k: v.float().cpu() for k, v in bench.random_state_dict(H, K, dev, gen, siglip_px=px).items()}
    cfg = bench.model_cfg(H, K, T)
    cfg.update(max_num_segments=mns, siglip_heads=16, dino_heads=24, qformer_heads=12)
    encs = {dt: VideoEncoder(sd, cfg, dtype=dt, device=dev) for dt in (torch.float16, torch.bfloat16)}
    vs = bench.synth_video(0, T, px, dev, torch.float32, scene_len=3)
    vd = bench.synth_video(0, T, px, dev, torch.float32, seed=4321, scene_len=3)
    ids = torch.tensor([[1, 2, 3, -200, 4, 5]])
    W = dict(sd)
    g = torch.Generator().manual_seed(3)
    table = torch.randn(64, H, generator=g) * 0.02
    W["embed_tokens_fn"] = lambda i: table[torch.as_tensor(i, dtype=torch.long) % 64]
    with torch.no_grad():
        r = orc.encode_video(W, cfg, vs.half().float().cpu(), vd.half().float().cpu(), (336, 336), ids, torch.tensor(PROMPT))
    want = r["visual_tokens"]
    for ta in (torch.float16, torch.bfloat16):
        for tb in (torch.float16, torch.bfloat16):
            keep = {}
            got = encode_video_with(Mixed(encs[ta], encs[tb]), vs.half().float(), vd.half().float(), (336, 336), ids.shape[1], ids.shape[1] - 1,
                                    PROMPT, keep=keep)
            plan = keep["plan"]
            comp = [i for i, e in enumerate(plan["src"]) if e[0] == "c"]
            stat = [i for i, e in enumerate(plan["src"]) if e[0] == "f"]
            d = (got.float().cpu() - want)
            ec, es = float(d[comp].abs().max()), float(d[stat].abs().max() / want[stat].abs().max())
            rms = float(d[comp].pow(2).mean().sqrt() / want[comp].pow(2).mean().sqrt())
            def rel(a, b):
                return float((a.float().cpu() - b).abs().max() / b.abs().max())
            print("towers %-8s rest %-8s | seg %s | siglip %.2e dino %.2e mm_proj %.2e | static rel %.2e | compressed max abs "
                  "%.2e rel rms %.2e" % (str(ta)[6:], str(tb)[6:], keep["seg_indices"] == [int(i) for i in r["seg_indices"]],
                                         rel(keep["siglip_feat"][:, :1152].reshape(T, 576, 1152), r["siglip_feat"]),
                                         rel(keep["dino_feat"][:, :1536].reshape(T, 576, 1536), r["dino_feat"]),
                                         rel(keep["mm_proj"][:, :H].reshape(T, 144, H), r["mm_proj"]), es, ec, rms), flush=True)


if __name__ == "__main__":
    main()
