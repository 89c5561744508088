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
