"""Gate of round 6's one experiment: fp16 tower operands over the fp16 residual stream with the LayerNorms folded into the
GEMMs around them and NO LayerNorm kernel inside the layer loop (the consumer GEMMs read the stream itself), against the
bench type (bf16 operands, fp16 stream, ln16_kernel passes) and against fp16 / fp16 without the fold.  Both towers at full depth,
`frames` frames in one batch; per configuration: tower time (events around tdc_vit_fwd), the library profiler's split, and the
features' distance from the fp16 / fp32-stream form (the closest this library has to the oracle).
    python tools/bench_ln_fold16.py [frames=512]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import ops  # noqa: E402
from tdc_video_amd.pipeline import VideoEncoder  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    H, K = 3584, 144
    sd = bench.random_state_dict(H, K, dev, gen)
    vs = bench.synth_video(0, T, 384, dev, torch.float16)
    vd = bench.synth_video(0, T, 378, dev, torch.float16, seed=4321)
    ref = {}
    rows = []
    for name, kw in (("fp16 / fp32 stream (reference form)", dict(tower_dtype=torch.float16, tower_res_dtype=None)),
                     ("bf16 / fp16 stream (bench type)", dict(tower_dtype=torch.bfloat16, tower_res_dtype=torch.float16)),
                     ("fp16 / fp16 stream", dict(tower_dtype=torch.float16, tower_res_dtype=torch.float16)),
                     ("fp16 / fp16 stream, LayerNorms folded", dict(tower_dtype=torch.float16, tower_res_dtype=torch.float16, ln_fuse=True))):
        enc = VideoEncoder(sd, bench.model_cfg(H, K, T), dtype=torch.float16, device=dev, tower_batch=T, **kw)
        out = {}
        for tower, px in (("siglip", vs), ("dino", vd)):
            enc.tower(tower, px)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                f = enc.tower(tower, px)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            ops.profile_start()
            enc.tower(tower, px)
            torch.cuda.synchronize()
            recs = ops.profile_stop()
            split = {k: sum(r["ms"] for r in recs if r["kind"] == k) for k in ("gemm", "attn", "ln")}
            f = f.float()
            if tower not in ref:
                ref[tower] = f
                err = 0.0
            else:
                err = float((f - ref[tower]).abs().max() / ref[tower].abs().max())
            out[tower] = (ms, split, err)
        tot = out["siglip"][0] + out["dino"][0]
        rows.append((name, tot))
        print("%-42s towers %8.2f ms | siglip %7.2f (gemm %.1f attn %.1f ln %.1f; vs ref %.2e) | dino %7.2f (gemm %.1f attn %.1f ln "
              "%.1f; vs ref %.2e)" % (name, tot, out["siglip"][0], out["siglip"][1]["gemm"], out["siglip"][1]["attn"],
                                      out["siglip"][1]["ln"], out["siglip"][2], out["dino"][0], out["dino"][1]["gemm"],
                                      out["dino"][1]["attn"], out["dino"][1]["ln"], out["dino"][2]), flush=True)
        del enc
        torch.cuda.empty_cache()
    base = rows[1][1]
    for name, tot in rows:
        print("%-42s %+.2f %% tower time against the bench type" % (name, (tot / base - 1) * 100))


if __name__ == "__main__":
    main()
