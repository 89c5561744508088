"""ORACLE (test infrastructure only) - CPU restatement of the reference's frame pre-processing, SURVEY.md 8(f)-3:
`process_images` (tdc/mm_datautils.py:286-314): per tower  expand2square(mean colour) (:270-282) ->
PIL `Image.resize((R, R))` (default resample for RGB = BICUBIC, with PIL's antialiasing support scaling) ->
HF image processor `preprocess` (resize / center-crop are no-ops at that point; rescale 1/255; normalize) -> fp16.

The resize lives in a third-party dependency (Pillow, pinned 10.4.0 in requirements.txt:106; 12.2.0 in this image):
restated here from its published algorithm (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
ImagingResampleHorizontal_8bpc / Vertical_8bpc: 22-bit fixed point, horizontal pass then vertical pass, uint8
intermediate) and PINNED in tests/test_preprocess.py against PIL itself (byte-exact) and against a fixture produced by
running the reference's own functions.  rescale / normalize follow transformers 4.46 image_transforms.py
(rescale: float64 multiply then cast to float32; normalize: (x - mean) / std in float32)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size, support=2.0):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for box (0, in_size): (bounds [out,2], coeffs int32
    [out, ksize])."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - sup + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + sup + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_axis(img, out_size, axis):
    """one 8bpc pass along `axis` (0 = vertical, 1 = horizontal) of img [H, W, C] uint8."""
    in_size = img.shape[axis]
    bounds, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)          # [in, other, C]
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def pil_resize_bicubic(img, out_w, out_h):
    """Image.resize((out_w, out_h), BICUBIC) on an RGB uint8 array [H, W, 3]: horizontal pass first, then vertical;
    a pass is skipped when that dimension does not change."""
    H, W = img.shape[:2]
    x = img
    if out_w != W:
        x = resample_axis(x, out_w, 1)
    if out_h != H:
        x = resample_axis(x, out_h, 0)
    return x


def expand2square(img, color):
    """tdc/mm_datautils.py:270-282."""
    H, W = img.shape[:2]
    if W == H:
        return img
    S = max(H, W)
    out = np.empty((S, S, 3), dtype=np.uint8)
    out[:] = np.array(color, dtype=np.uint8)
    if W > H:
        y0 = (W - H) // 2
        out[y0:y0 + H] = img
    else:
        x0 = (H - W) // 2
        out[:, x0:x0 + W] = img
    return out


def process_frames(frames, R, image_mean, image_std):
    """frames uint8 [T, H, W, 3] -> fp16 [T, 3, R, R] (one tower)."""
    color = tuple(int(x * 255) for x in image_mean)
    mean = np.array(image_mean, dtype=np.float32)
    std = np.array(image_std, dtype=np.float32)
    out = []
    for f in frames:
        sq = expand2square(f, color)
        rs = pil_resize_bicubic(sq, R, R)
        x = (rs.astype(np.float64) * (1.0 / 255.0)).astype(np.float32)
        x = (x - mean) / std
        out.append(np.transpose(x, (2, 0, 1)))
    return np.stack(out).astype(np.float16)
