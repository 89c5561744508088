"""Print per-stage relative errors of the HIP path vs the golden fixtures (calibration aid for the test tolerances)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import synth  # noqa: E402
from util import oracle, load_fixture, embed_fn, pipeline_cfg  # noqa: E402
import tdc_video_amd  # noqa: F401,E402
from tdc_video_amd import weights as Wt  # noqa: E402
from tdc_video_amd.pipeline import VideoEncoder  # noqa: E402


def rel(a, b):
    a, b = a.float().cpu(), torch.as_tensor(b).float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


for dtype in (torch.float16, torch.bfloat16):
    print("=====", dtype)
    for name, prep in (("siglip_small.npz", "siglip"), ("dino_small.npz", "dino")):
        W, o = load_fixture(name)
        enc = VideoEncoder.__new__(VideoEncoder)
        enc.dtype, enc.dev, enc.tower_batch = dtype, torch.device("cuda"), 64
        enc._tables = {}
        enc.out_grid = [8, 8]
        t = (Wt.prep_siglip if prep == "siglip" else Wt.prep_dino)(W, 4, dtype, enc.dev)
        enc.towers = {prep: t}
        px = torch.from_numpy(o["pixels"]).cuda()
        out = enc.tower(prep, px)
        print(name, "tower rel err", rel(out[:, :t.dim].reshape(px.shape[0], 64, t.dim), o["out"]))
    for name in ("pipeline_T40.npz", "pipeline_T10_land.npz", "pipeline_T260.npz"):
        W, o = load_fixture(name)
        cfg = pipeline_cfg(o)
        enc = VideoEncoder(W, cfg, dtype=dtype, device="cuda", siglip_heads=4, dino_heads=4, qformer_heads=4)
        vid = torch.from_numpy(synth.video_from_basis(o["video_basis"], o["video_coef"]))
        ids = torch.from_numpy(o["input_ids"])[0]
        size = tuple(int(v) for v in o["image_size"])
        keep = {}
        vis = enc.encode_video(vid.cuda(), (vid + 0.01).cuda(), size, len(ids), len(ids) - 1,
                               [int(i) for i in o["prompt_ids"]], keep=keep)
        T = len(keep["selected"])
        st = 16 if "T260" in name else 1
        print(name, "seg ok", keep["seg_indices"] == o["out_seg_indices"].tolist(), "T", T, "vis", tuple(vis.shape))
        for key, ref, cols in (("siglip_feat", "out_siglip_feat", 48), ("dino_feat", "out_dino_feat", 64),
                               ("aux0", "out_aux0", 64), ("aux1", "out_aux1", 64), ("sva", "out_sva", 64),
                               ("mm_proj", "out_mm_proj", 96)):
            g = keep[key][:, :cols].reshape(T, -1, cols)[::st]
            print("   %-12s rel %.3e   (max|ref| %.2f)" % (key, rel(g, o[ref]), abs(o[ref]).max()))
        emb = embed_fn(o)
        pos = int(torch.where(ids == -200)[0][0])
        full = torch.cat([emb(ids[:pos]), vis.float().cpu(), emb(ids[pos + 1:])])[: cfg["tokenizer_model_max_length"]]
        ref = torch.from_numpy(o["out_inputs_embeds"])[0]
        print("   inputs_embeds shape", tuple(full.shape), tuple(ref.shape), "rel", rel(full, ref) if full.shape == ref.shape else None)
        if "compressed" in keep:
            W["embed_tokens_fn"] = emb
            r = oracle.encode_video(W, cfg, vid, vid + 0.01, size, torch.from_numpy(o["input_ids"]),
                                    torch.from_numpy(o["prompt_ids"]))
            plan = keep["plan"]
            comp = keep["compressed"][:, :96].float().cpu()
            errs = [(comp[e[1] * 4 + e[2]] - r["visual_tokens"][i]).abs().max().item()
                    for i, e in enumerate(plan["src"]) if e[0] == "c"]
            print("   compressed tokens abs err max %.3e  (n=%d)" % (max(errs), len(errs)))
