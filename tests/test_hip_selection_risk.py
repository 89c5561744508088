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


@pytest.fixture(scope="module")
def world():
    return build_world()


def build_world():
    """the bench-type encoder at full size (both towers: the second test runs the whole path) + the slow-drift video"""
    import bench
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd import weights as Wt
    from tdc_video_amd.pipeline import VideoEncoder
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    H, K = 3584, 144
    sd = bench.random_state_dict(H, K, dev, gen)
    d_sd = {k[len("vision_tower_aux_list.1.vision_tower."):]: v for k, v in sd.items() if k.startswith("vision_tower_aux_list.1.")}
    enc = VideoEncoder(sd, bench.model_cfg(H, K, T), dtype=torch.float16, device=dev, tower_batch=512, tower_dtype=torch.bfloat16,
                       tower_res_dtype=torch.float16)
    del sd
    torch.cuda.empty_cache()

    def sims_of(video, operand, stream):
        """similarities of `video` through a DINOv2 tower of the given operand / stream types (the engine's own tower restored)"""
        keep = enc.towers["dino"], enc.tower_res_dtype
        t = Wt.prep_dino(d_sd, 24, operand, dev)
        t["dtype"] = operand
        enc.towers["dino"] = t
        enc.tower_res_dtype = stream
        try:
            return enc.sims_tensor(enc.tower("dino", video), video.shape[0]).cpu().tolist()
        finally:
            enc.towers["dino"], enc.tower_res_dtype = keep

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
    return enc, sims_of, _video(A, B, steps)


def test_selection_risk_of_the_dino_operand_type(world):
    from tdc_video_amd import segment as seg
    enc, sims_of, video = world
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


def test_refinement_gives_the_bench_type_the_fp16_selection(world):
    """VideoEncoder(selection_refine) - automatic under bf16 DINOv2 operands: when the similarity ranks that decide the a5 selection
    are closer than 4 selection_eps, the pairs inside the band around them, and only they, are re-encoded by the fp16-operand copy of the
    tower and re-ranked.  On the slow-drift video, for EVERY boundary rank 4 ... 200: the refined selection of the bench type is
    exactly what ranking the fp16 / fp16 tower's similarities selects (the reference's own arithmetic, tdc/builder.py:69), at the
    price of a few dozen re-encoded frames; and the whole path (encode_video) does it by itself."""
    import bench
    from tdc_video_amd import segment as seg
    enc, sims_of, video = world
    assert enc.selection_eps == 1e-3 and "dino_precise" in enc.towers and enc.towers["dino_precise"]["dtype"] == torch.float16
    feat_b = enc.tower("dino", video)
    s_b = enc.sims_tensor(feat_b, T).cpu().tolist()
    feat_p = enc.precise_dino(video)
    s_p = enc.sims_tensor(feat_p, T).cpu().tolist()
    assert s_p == sims_of(video, torch.float16, torch.float16)                       # the precise tower IS the fp16 / fp16 form
    err = max(abs(a - b) for a, b in zip(s_b, s_p))
    assert err < 0.6 * enc.selection_eps, err                                        # measured 3.7e-4 against the assumed 1e-3
    # the bound the refinement rests on, on other kinds of video: scenes with cuts (the bench's), independent noise frames
    # (similarities near 0.5), a near-static clip (similarities near 1)
    gg = torch.Generator(device=video.device).manual_seed(3)
    base = torch.rand(3, 378, 378, device=video.device, generator=gg) * 2 - 1
    others = {"scenes + cuts (bench video)": bench.synth_video(0, 128, 378, video.device, torch.bfloat16, seed=4321),
              "independent noise frames": (torch.rand(128, 3, 378, 378, device=video.device, generator=gg) * 2 - 1).half(),
              "near-static clip": (base[None] + 0.02 * torch.randn(128, 3, 378, 378, device=video.device, generator=gg)).half()}
    for name, v in others.items():
        sb = enc.sims_tensor(enc.tower("dino", v), v.shape[0]).cpu()
        sp = enc.sims_tensor(enc.precise_dino(v), v.shape[0]).cpu()
        e = float((sb - sp).abs().max())
        print("similarity error of the bench type on %-30s %.2e (similarities %.3f ... %.3f)" % (name + ":", e, float(sp.min()), float(sp.max())))
        assert e < 0.6 * enc.selection_eps, (name, e)
    P = feat_p.shape[0] // T
    rows = {f: feat_p[f * P:(f + 1) * P] for f in range(T)}
    sizes, changed = [], 0
    for m in range(4, 201):
        band = seg.selection_band(s_b, m, enc.selection_eps)
        refined = enc.pair_sims(rows, [(i, i + 1) for i in band]).tolist() if band else []
        assert refined == [s_p[i] for i in band]                                     # pair_sims == the a5 kernel on the same rows
        got = seg.select_refined(s_b, m, enc.selection_eps, band, refined)
        assert got == seg.select_segments(s_p, m), m
        changed += got != seg.select_segments(s_b, m)
        sizes.append(len(seg.band_frames(band)))
    print("refinement on the slow-drift video: similarity error of the bench type %.2e (bound %.0e); boundary ranks 4..200: the "
          "refined selection equals the fp16 tower's at all 197, differs from the unrefined one at %d; frames re-encoded per call: "
          "median %d, max %d of %d" % (err, enc.selection_eps, changed, int(np.median(sizes)), max(sizes), T))
    assert changed >= 10 and max(sizes) <= T // 8           # inside the default cost cap (selection_max_fraction = 1 / 8)
    # the re-encoded subset equals the rows of the full run bit for bit (tower-batch invariance), and encode_video runs it by itself
    band = seg.selection_band(s_b, 24, enc.selection_eps)
    frames = seg.band_frames(band)
    sub = enc.precise_dino(video[torch.tensor(frames, device=video.device)])
    assert torch.equal(sub, torch.cat([rows[f] for f in frames], 0))
    vs = bench.synth_video(0, T, 384, video.device, torch.bfloat16)
    info = {}
    out = enc.encode_video(vs, video, (384, 384), budget_text_len=64, n_text_tokens=64, prompt_ids=[101, 2000, 102], frame_cap=T,
                           info=info)
    assert info["refined_pairs"] == band and len(band) >= 2
    assert info["seg_indices"] == seg.select_segments(s_p, 24)
    # the cost cap: with room for 8 frames only the band (a plateau, as far as the cap is concerned) is left to the fast ranking
    enc.selection_max_fraction = 8.0 / T / 2
    info3 = {}
    enc.encode_video(vs, video, (384, 384), budget_text_len=64, n_text_tokens=64, prompt_ids=[101, 2000, 102], frame_cap=T, info=info3)
    enc.selection_max_fraction = 0.125
    assert info3["refined_pairs"] == [] and info3["refine_skipped_pairs"] == len(band) and info3["seg_indices"] == seg.select_segments(s_b, 24)
    enc.selection_eps = None                                                         # refinement off: the bench type's own ranking
    info2 = {}
    out2 = enc.encode_video(vs, video, (384, 384), budget_text_len=64, n_text_tokens=64, prompt_ids=[101, 2000, 102], frame_cap=T,
                            info=info2)
    enc.selection_eps = 1e-3
    assert info2["refined_pairs"] == [] and info2["seg_indices"] == seg.select_segments(s_b, 24)
    assert torch.isfinite(out.float()).all() and (info["seg_indices"] == info2["seg_indices"]) == torch.equal(out, out2)
