"""GPU (-m gpu): how much of the a5 segment selection (tdc/cambrian_arch.py:832-849: the 24 lowest adjacent-frame cosine
similarities of the DINOv2 features) is at risk from the operand type of the DINOv2 tower.

The other full-depth tests plant a ranking margin of 0.43; real video does not: the 24th and 25th smallest similarities of a
long clip are routinely < 1e-3 apart.  Here a 512-frame "slow drift" video - f_t = cos(theta_t) A + sin(theta_t) B with angle
steps chosen (after one calibration pass) so that the 511 similarities cover an interval of ~0.25 about uniformly, i.e. the gaps
between neighbouring ranks are 1e-4 ... 1e-3 around EVERY candidate boundary - goes through the 40-layer DINOv2 tower in three
arithmetic types:
    fp16 operands / fp32 residual stream   the closest this library has to the fp32 oracle (similarities 1.9e-5)
    fp16 operands / fp16 residual stream   the reference's own inference arithmetic (tdc/builder.py:69)
    bf16 operands / fp16 residual stream   bench.py's type
Per type, against the first: the similarity error; the number of `seg_indices` that differ at max_num_segments = 24 and over every
boundary rank 4 ... 200; and the SMALLEST SAFE GAP - the largest gap (in the reference similarities) between two frame pairs that
the type ranks the other way round.  A boundary whose two neighbours are further apart than that is selected identically.
`config.tdc_dino_dtype = "float16"` (model.tdc_engine; VideoEncoder(dino_dtype=...)) gives a caller the fp16 row of this table
for the DINOv2 tower alone while SigLIP keeps bf16 operands.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

T = 512


def _video(A, B, steps):
    th = torch.cat([torch.zeros(1, device=A.device), torch.cumsum(steps, 0)])
    return (torch.cos(th)[:, None, None, None] * A[None] + torch.sin(th)[:, None, None, None] * B[None]).half()


def _inversions(ref, got):
    """largest ref-gap between two pairs that `got` ranks the other way round (ties in `got` count as kept)"""
    r = np.asarray(ref, dtype=np.float64)
    g = np.asarray(got, dtype=np.float64)
    dr = r[:, None] - r[None, :]
    dg = g[:, None] - g[None, :]
    bad = (dr > 0) & (dg < 0)
    return float(dr[bad].max()) if bad.any() else 0.0, int(bad.sum())


def test_selection_risk_of_the_dino_operand_type():
    import bench
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd import segment as seg
    from tdc_video_amd import weights as Wt
    from tdc_video_amd.pipeline import VideoEncoder
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    H, K = 3584, 144
    sd = bench.random_state_dict(H, K, dev, gen)
    d_sd = {k[len("vision_tower_aux_list.1.vision_tower."):]: v for k, v in sd.items() if k.startswith("vision_tower_aux_list.1.")}
    sd = {k: v for k, v in sd.items() if not k.startswith("vision_tower_aux_list.0.")}       # DINOv2 + connector: a5 needs no SigLIP
    enc = VideoEncoder(sd, bench.model_cfg(H, K, T), dtype=torch.float16, device=dev, tower_batch=512)
    del sd
    torch.cuda.empty_cache()

    def sims_of(video, operand, stream):
        t = Wt.prep_dino(d_sd, 24, operand, dev)
        t["dtype"] = operand
        enc.towers["dino"] = t
        enc.tower_res_dtype = stream
        s = enc.sims_tensor(enc.tower("dino", video), video.shape[0]).cpu().tolist()
        del enc.towers["dino"]
        return s

    g = torch.Generator(device=dev).manual_seed(77)
    A = torch.rand(3, 378, 378, device=dev, generator=g) * 2 - 1
    B = torch.rand(3, 378, 378, device=dev, generator=g) * 2 - 1
    # calibration pass (reference type): 1 - similarity against the angle step, on a geometric ladder of steps
    ladder = torch.tensor(np.geomspace(2e-3, 0.6, T - 1), device=dev, dtype=torch.float32)
    q = 1.0 - np.asarray(sims_of(_video(A, B, ladder), torch.float16, None))
    q_mono = np.maximum.accumulate(q)
    lo, hi = 0.02, 0.27
    assert q_mono[0] < lo and q_mono[-1] > hi, (q_mono[0], q_mono[-1])       # the ladder brackets the target interval
    targets = lo + (hi - lo) * (np.arange(T - 1) + 0.5) / (T - 1)
    steps = np.interp(targets, q_mono, ladder.cpu().numpy())
    rng = np.random.RandomState(5)
    steps = torch.tensor(steps[rng.permutation(T - 1)], device=dev, dtype=torch.float32)
    video = _video(A, B, steps)

    ref = sims_of(video, torch.float16, None)
    srt = np.sort(np.asarray(ref))
    gaps = np.diff(srt)
    print("reference similarities: min %.4f max %.4f; gaps between ranks 20..30: %s; median gap %.2e"
          % (srt[0], srt[-1], " ".join("%.1e" % v for v in gaps[19:30]), float(np.median(gaps))))
    assert 5e-5 < float(np.median(gaps)) < 1.5e-3 and float(gaps[19:30].max()) < 5e-3        # the video does what it says
    ref_sel = {m: seg.select_segments(ref, m) for m in range(4, 201)}
    rows = {}
    for name, operand, stream in (("fp16 / fp16", torch.float16, torch.float16), ("bf16 / fp16 (bench)", torch.bfloat16, torch.float16)):
        s = sims_of(video, operand, stream)
        err = float(np.abs(np.asarray(s) - np.asarray(ref)).max())
        safe_gap, n_inv = _inversions(ref, s)
        diff24 = len(set(seg.select_segments(s, 24)) ^ set(ref_sel[24])) // 2
        diffs = [len(set(seg.select_segments(s, m)) ^ set(ref_sel[m])) // 2 for m in range(4, 201)]
        # every boundary whose two neighbouring ranks are further apart than safe_gap is selected identically
        for m in range(4, 201):
            if srt[m] - srt[m - 1] > safe_gap:
                assert seg.select_segments(s, m) == ref_sel[m], (name, m)
        rows[name] = (err, safe_gap, n_inv, diff24, sum(1 for d in diffs if d), max(diffs))
        print("%-22s similarity max abs err %.2e; smallest safe gap %.2e (%d inverted pairs of %d); seg_indices differing at "
              "max_num_segments = 24: %d; boundaries 4..200 with a different selection: %d of 197 (at most %d indices)"
              % ((name, err, safe_gap, n_inv, (T - 1) * (T - 2) // 2, diff24) + rows[name][4:]))
    e16, g16 = rows["fp16 / fp16"][:2]
    eb, gb = rows["bf16 / fp16 (bench)"][:2]
    # measured (profiles/r06_selection_risk.log): fp16 / fp16 2.87e-5 / 1.21e-5 (6 inverted pairs of 130 305, 2 of 197 boundaries
    # move one index); bench type 3.65e-4 / 3.06e-4 (52 pairs, 32 of 197 boundaries move one index).  Bounds <= 1.5 x measured,
    # except the fp16 gap, asserted in the form it is claimed in: nothing differs above a 1e-4 gap
    assert e16 < 4.5e-5 and g16 < 1.0e-4, (e16, g16)
    assert eb < 5.5e-4 and gb < 4.6e-4, (eb, gb)
    assert g16 < gb                                            # what config.tdc_dino_dtype = "float16" buys
