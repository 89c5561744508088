"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the oracle."""
import torch

import synth
from util import oracle, load_fixture, embed_fn, pipeline_cfg


def run():
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd.pipeline import VideoEncoder
    W, o = load_fixture("pipeline_T40.npz")
    cfg = pipeline_cfg(o)
    enc = VideoEncoder(W, cfg, dtype=torch.float16, device="cuda:0", siglip_heads=4, dino_heads=4, qformer_heads=4)
    vid = torch.from_numpy(synth.video_from_basis(o["video_basis"], o["video_coef"]))
    ids = torch.from_numpy(o["input_ids"])[0]
    size = tuple(int(v) for v in o["image_size"])
    vis = enc.encode_video(vid.cuda(), (vid + 0.01).cuda(), size, len(ids), len(ids) - 1,
                           [int(i) for i in o["prompt_ids"]])
    torch.cuda.synchronize()
    W["embed_tokens_fn"] = embed_fn(o)
    r = oracle.encode_video(W, cfg, vid, vid + 0.01, size, torch.from_numpy(o["input_ids"]),
                            torch.from_numpy(o["prompt_ids"]))
    ref = r["visual_tokens"]
    assert vis.shape == ref.shape, (vis.shape, ref.shape)
    err = ((vis.float().cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < 4e-3, err
    print("smoke ok: %d visual tokens, rel err vs oracle %.2e" % (vis.shape[0], err))
