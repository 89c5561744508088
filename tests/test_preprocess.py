"""SURVEY 8(f)-3, frame pre-processing.  CPU: the oracle (numpy restatement of Pillow's bicubic resample + the HF
rescale/normalize) against Pillow itself (byte-exact) and against the fixture produced by running the reference's own
process_images; the product's host tables against the oracle.  GPU: the HIP kernels bit-exact against the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import preprocess_oracle as po  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "preprocess_small.npz")


def test_oracle_resize_matches_pillow_bytes():
    from PIL import Image
    rng = np.random.RandomState(0)
    for (H, W, R) in [(45, 80, 42), (80, 45, 56), (360, 640, 384), (64, 64, 42), (100, 37, 56), (30, 50, 64)]:
        img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
        img[: H // 3] = 255                       # saturated regions exercise the clipping of bicubic overshoot
        ref = np.asarray(Image.fromarray(img).resize((R, R)))
        assert np.array_equal(po.pil_resize_bicubic(img, R, R), ref), (H, W, R)


def test_oracle_matches_reference_process_images():
    z = np.load(GOLD)
    for tag in ("land", "port", "square"):
        for tw, R in (("siglip", 42), ("dino", 56)):
            got = po.process_frames(z["frames_" + tag], R, z[tw + "_mean"], z[tw + "_std"])
            assert np.array_equal(got, z["out_%s_%s" % (tw, tag)]), (tag, tw)      # fp16, bit-exact


def test_product_tables_match_oracle():
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd import preprocess as pp
    for (a, b) in [(80, 42), (640, 384), (1920, 378), (50, 56), (300, 300 + 77)]:
        bo, ko = po.precompute_coeffs(a, b)
        bp, kp = pp.resize_tables(a, b)
        assert np.array_equal(bo, bp) and np.array_equal(ko, kp)
    lut = pp.norm_table((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    v = np.arange(256, dtype=np.uint8).reshape(1, 256, 1, 1).repeat(3, axis=3)
    want = po.process_frames(np.broadcast_to(v, (1, 256, 256, 3)).copy()[:, :, :1].repeat(256, axis=2), 256,
                             (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    assert np.array_equal(lut.astype(np.float16)[:, :], want[0, :, :, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_hip_preprocess_bit_exact(dtype):
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd import preprocess as pp
    z = np.load(GOLD)
    rng = np.random.RandomState(3)
    cases = [(z["frames_land"], 42), (z["frames_port"], 56), (z["frames_square"], 42), (z["frames_square"], 50),
             (rng.randint(0, 256, (2, 360, 640, 3)).astype(np.uint8), 384),
             (rng.randint(0, 256, (2, 378, 378, 3)).astype(np.uint8), 378)]
    for frames, R in cases:
        for tw in (pp.SIGLIP, pp.DINOV2):
            got = pp.preprocess_frames(torch.from_numpy(frames), R, tw["mean"], tw["std"], dtype, out_f32=True)
            want32 = po.process_frames.__globals__["np"].stack  # noqa: F841 (keep numpy handle)
            ref16 = po.process_frames(frames, R, tw["mean"], tw["std"])            # fp16 reference
            g16 = pp.preprocess_frames(torch.from_numpy(frames), R, tw["mean"], tw["std"], torch.float16)
            assert np.array_equal(g16.cpu().numpy(), ref16), (frames.shape, R)
            if dtype == torch.bfloat16:
                gb = pp.preprocess_frames(torch.from_numpy(frames), R, tw["mean"], tw["std"], torch.bfloat16)
                assert torch.equal(gb.cpu(), got.cpu().to(torch.bfloat16))          # same fp32 values, RNE to bf16


@pytest.mark.gpu
def test_hip_process_images_matches_reference_fixture():
    import tdc_video_amd  # noqa: F401
    from tdc_video_amd import preprocess as pp
    z = np.load(GOLD)
    towers = (dict(R=42, mean=tuple(z["siglip_mean"]), std=tuple(z["siglip_std"])),
              dict(R=56, mean=tuple(z["dino_mean"]), std=tuple(z["dino_std"])))
    for tag in ("land", "port", "square"):
        out = pp.process_images(list(z["frames_" + tag]), torch.float16, towers)
        assert np.array_equal(out[0].cpu().numpy(), z["out_siglip_" + tag])
        assert np.array_equal(out[1].cpu().numpy(), z["out_dino_" + tag])
