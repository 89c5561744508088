"""Deterministic synthetic video used by the golden fixtures AND by the tests that replay them.

A video is `coef [T, NB] @ basis [NB, 3*px*px]` (fp32): only the small fp16 basis and the coefficient matrix are
stored in the fixture, the frames are rebuilt with the same formula on both sides.
"""
import numpy as np


def make_basis_and_coef(T, px, nb=8, seed=0, scene_len=3):
    rng = np.random.RandomState(seed)
    basis = (rng.rand(nb, 3, px, px).astype(np.float32) * 2 - 1).astype(np.float16)
    coef = np.zeros((T, nb), dtype=np.float32)
    scene = 0
    for t in range(T):
        if t > 0 and t % scene_len == 0:
            scene = (scene + 1 + rng.randint(0, nb - 1)) % nb
        coef[t, scene] = 1.0
        coef[t] += (0.02 * (t % scene_len + 1)) * rng.randn(nb).astype(np.float32)
    return basis, coef


def video_from_basis(basis, coef):
    nb = basis.shape[0]
    flat = basis.astype(np.float32).reshape(nb, -1)
    vid = coef.astype(np.float32) @ flat
    return vid.reshape((coef.shape[0],) + basis.shape[1:]).astype(np.float32)


class FakeBeats:
    """Stand-in for BEATs.extract_features (tdc/audio_models/beats/BEATs.py:131-178; needs torchaudio, absent in the
    build container): deterministic [1, n, 768] features with n = floor(seconds * 49.6) - a full 10-s window gives 496
    frames, so its last second has 46 tokens and takes the reference's adaptive_avg_pool2d branch
    (tdc/cambrian_arch.py:1567-1568).  Used by make_golden.py (plugged into the reference) and by the tests."""

    def extract_features(self, wav, padding_mask=None, feature_only=True):
        import torch
        n = int(wav.shape[1] / 16000.0 * 49.6)
        seg = wav[0, : n * 320].reshape(n, 320)
        base = seg.mean(1, keepdim=True) * 50.0 + seg.std(1, keepdim=True)
        emb = torch.sin(base * torch.arange(1, 769).float()[None] * 0.37) + base
        return emb[None], None


def beats_windows(wav, dist=10, sr=16000):
    """BEATs features of the consecutive 10-s windows of wav [1, n] (tdc/cambrian_arch.py:1552-1560)."""
    out = []
    for k in range(0, int(wav.shape[1] / sr), dist):
        out.append(FakeBeats().extract_features(wav[:, sr * k: sr * (k + dist)])[0])
    return out
