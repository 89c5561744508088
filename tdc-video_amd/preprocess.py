"""Frame pre-processing on the device: drop-in for the reference's `process_images` (tdc/mm_datautils.py:286-314) on
video frames - uint8 HWC frames in, one fp16/bf16 [T,3,R,R] tensor per tower out (SigLIP 384 / mean 0.5, DINOv2 378 /
ImageNet mean).  The host only builds the small resampling tables (Pillow's Resample.c precompute_coeffs, done in
double precision exactly like the C code) and the 3x256 normalisation table; the pixels never leave the GPU."""
import math

import numpy as np
import torch

from . import lib as L
from . import ops

PRECISION_BITS = 32 - 8 - 2

SIGLIP = dict(R=384, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5))
DINOV2 = dict(R=378, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225))


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resize_tables(in_size, out_size, support=2.0):
    """Pillow src/libImaging/Resample.c: precompute_coeffs (box = whole axis) + normalize_coeffs_8bpc."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - sup + 0.5), 0)
        xmax = min(int(center + sup + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def norm_table(mean, std):
    """lut[c, v] = ((v * 1/255 in float64 -> float32) - mean32[c]) / std32[c]   (HF rescale + normalize)."""
    v = (np.arange(256, dtype=np.float64) * (1.0 / 255.0)).astype(np.float32)
    m = np.array(mean, dtype=np.float32)
    s = np.array(std, dtype=np.float32)
    return ((v[None, :] - m[:, None]) / s[:, None]).astype(np.float32)


_cache = {}


def preprocess_frames(frames, R, mean, std, dtype=torch.float16, out_f32=False):
    """frames: uint8 tensor [T, H, W, 3] (cpu or cuda) -> [T, 3, R, R] on the GPU."""
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3
    frames = frames.cuda().contiguous() if not frames.is_cuda else frames.contiguous()
    T, H, W, _ = frames.shape
    S = max(H, W)
    key = (S, R, tuple(mean), tuple(std), frames.device)
    if key not in _cache:
        b, k = resize_tables(S, R) if S != R else (np.zeros((1, 2), np.int32), np.zeros((1, 1), np.int32))
        _cache[key] = (torch.from_numpy(b).to(frames.device), torch.from_numpy(k).to(frames.device).contiguous(),
                       torch.from_numpy(norm_table(mean, std)).to(frames.device).contiguous())
    bounds, coeffs, lut = _cache[key]
    lib = L.load()
    out = torch.empty(T, 3, R, R, device=frames.device, dtype=torch.float32 if out_f32 else dtype)
    nscr = lib.tdc_preprocess_scratch_bytes(T, H, W, R)
    scratch = torch.empty(max(nscr, 1), dtype=torch.uint8, device=frames.device)
    pad = [int(x * 255) for x in mean]        # expand2square background (tdc/mm_datautils.py:302-304)
    L.check(lib.tdc_preprocess_frames(ops._ptr(frames), T, H, W, R, ops._ptr(bounds), ops._ptr(coeffs), coeffs.shape[1],
                                      pad[0], pad[1], pad[2], ops._ptr(lut), ops._ptr(out), int(out_f32),
                                      ops._dtcode(dtype), ops._ptr(scratch), ops._stream()), "tdc_preprocess_frames")
    return out


def process_images(frames, dtype=torch.float16, towers=(SIGLIP, DINOV2)):
    """Mirror of tdc/mm_datautils.py process_images for a list / array of video frames: returns one tensor per tower."""
    if not torch.is_tensor(frames):
        frames = torch.from_numpy(np.ascontiguousarray(np.stack(list(frames))))
    return [preprocess_frames(frames, t["R"], t["mean"], t["std"], dtype) for t in towers]
