"""BEATs audio encoder.  CPU part: the oracle restatement against the fixture generated from the
imported reference (tests/golden/make_golden_beats.py) and the known-answer anchors of the kaldi fbank restatement
(torchaudio is absent: that stage is "parity unpinned" by the reference - and cross-checked against the kaldi-compatible
implementation of transformers.audio_utils, an independent third party).  GPU part: tests/test_hip_beats.py."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from util import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import beats_oracle as BO  # noqa: E402


def load_beats_fixture():
    z = np.load(os.path.join(GOLDEN, "beats_small.npz"), allow_pickle=False)
    W = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("W/")}
    cfg = {}
    for k, v in zip(z["cfg_keys"], z["cfg_vals"]):
        v = str(v)
        cfg[str(k)] = (v == "True") if v in ("True", "False") else (float(v) if "." in v else (int(v) if v.lstrip("-").isdigit() else v))
    o = {k: z[k] for k in z.files if not k.startswith("W/")}
    return W, cfg, o


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_vs_reference_fixture(tag):
    W, cfg, o = load_beats_fixture()
    wav = torch.from_numpy(o["wav_" + tag].astype(np.float32))
    fb = BO.preprocess(wav)
    assert torch.equal(fb, torch.from_numpy(o["fbank_" + tag]))          # same restatement as the generator's shim
    out = BO.features_from_fbank(W, cfg, torch.from_numpy(o["fbank_" + tag]))
    ref = torch.from_numpy(o["out_" + tag])
    assert out.shape == ref.shape
    assert float((out - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    out2 = BO.extract_features(W, cfg, wav, padding_mask=torch.zeros(wav.shape, dtype=torch.bool))
    assert float((out2 - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_oracle_padding_mask_branch():
    W, cfg, o = load_beats_fixture()
    wav = torch.from_numpy(o["wav_pad"].astype(np.float32))
    mask = torch.from_numpy(o["mask_pad"])
    fbm = BO.forward_padding_mask(BO.preprocess(wav).shape[1], mask)
    tokm = BO.forward_padding_mask(o["out_pad"].shape[1], fbm)
    assert torch.equal(tokm, torch.from_numpy(o["tokmask_pad"]))
    out = BO.extract_features(W, cfg, wav, padding_mask=mask)
    ref = torch.from_numpy(o["out_pad"])
    valid = ~tokm
    assert float((out - ref)[valid].abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_relative_position_buckets_properties():
    R = 2000
    b = BO.relative_position_bucket(torch.arange(-R, R + 1)[None], 320, 800)[0]
    assert int(b.min()) == 0 and int(b.max()) == 319
    assert b[R] == 0 and b[R + 1] == 161 and b[R - 1] == 1 and b[R + 79] == 160 + 79 and b[R - 79] == 79
    assert b[R + 80] == 160 + 80 and b[0] == 159 and b[2 * R] == 319          # log region starts at 80, saturates
    assert torch.all(b[R + 1:] >= 160) and torch.all(b[:R] < 160)
    neg = b[:R + 1].flip(0)                                                     # |rel| = 0..R on the negative side
    assert torch.all(neg[1:] >= neg[:-1])
    assert torch.equal(b[R + 1:] - 160, neg[1:])                                # symmetric apart from the sign offset


def test_fbank_known_answers():
    """anchors for the unpinned kaldi fbank restatement."""
    n = 16000
    assert BO.kaldi_fbank(torch.zeros(399)).shape == (0, 128)
    assert BO.kaldi_fbank(torch.zeros(400)).shape == (1, 128)
    assert BO.kaldi_fbank(torch.zeros(160000)).shape == (998, 128)           # 10-s window -> 998 frames -> 62x8 patches
    sil = BO.kaldi_fbank(torch.zeros(n))
    assert torch.allclose(sil, torch.full_like(sil, math.log(torch.finfo(torch.float32).eps)))
    t = torch.arange(n) / 16000.0
    banks = BO.mel_banks()
    assert banks.shape == (128, 256) and float(banks.min()) >= 0 and float(banks.max()) <= 1.0
    for f0 in (440.0, 1000.0, 3000.0):
        fb = BO.kaldi_fbank(torch.sin(2 * math.pi * f0 * t) * 2 ** 14)
        peak = int(fb.mean(0).argmax())
        want = int(banks[:, int(round(f0 / 31.25))].argmax())
        assert abs(peak - want) <= 1, (f0, peak, want)
    x = torch.randn(n, generator=torch.Generator().manual_seed(0)) * 1000
    a, b = BO.kaldi_fbank(x), BO.kaldi_fbank(4 * x)
    live = a > -15.0                          # a few of the narrowest low mel bins contain no FFT bin: floor value
    assert int((~live).sum()) < 0.05 * a.numel()
    assert torch.allclose((b - a)[live], torch.full_like(a[live], math.log(16.0)), atol=1e-2)   # power scale law
    c = BO.kaldi_fbank(x + 500.0)                                                  # DC removal
    assert float((a - c).abs().max()) < 0.1


def test_kaldi_fbank_restatement_vs_transformers_kaldi_compat():
    """The kaldi fbank front end (tdc/audio_models/beats/BEATs.py:116-129 calls torchaudio.compliance.kaldi.fbank; torchaudio
    is absent from this image and the reference holds no vectors, so this stage cannot be pinned by the reference's own
    dependency) cross-checked against an INDEPENDENT third-party implementation that is installed: the kaldi-compatible path of
    `transformers.audio_utils` (povey window, pre-emphasis 0.97, DC removal, kaldi mel scale triangularised in mel space, log with
    the float32-epsilon floor) - the code HF's SeamlessM4T / AST feature extractors run in place of torchaudio's kaldi.fbank.
    It computes in float64; the restatement in float32 like torchaudio: agreement 2e-3 max / 1e-5 mean on log-mel values in
    [-16, 28] (measured 1.7e-3 / 8e-6), the same frame count, and the same mel filter bank."""
    au = pytest.importorskip("transformers.audio_utils")
    import warnings
    window = au.window_function(400, "povey", periodic=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # "at least one mel filter has all zero values": true of kaldi's 128 bins at 512 points too
        mel = au.mel_filter_bank(num_frequency_bins=257, num_mel_filters=128, min_frequency=20, max_frequency=8000,
                                 sampling_rate=16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    assert float(np.abs(mel.T[:, :256] - BO.mel_banks().numpy()[:, :256]).max()) < 5e-5     # fp32 (kaldi) vs float64 weights: 1.4e-5
    g = torch.Generator().manual_seed(1)
    for n, kind in [(16000 * 3 + 123, "chirp"), (16000 * 2, "noise"), (400, "one frame"), (16000 * 10, "noise")]:
        t = torch.arange(n) / 16000.0
        wav = (0.3 * torch.sin(2 * math.pi * (200 + 500 * t) * t) + 0.02 * torch.randn(n, generator=g)) if kind == "chirp" \
            else 0.1 * torch.randn(n, generator=g)
        mine = BO.kaldi_fbank(wav * 2 ** 15).numpy()
        ref = au.spectrogram(wav.numpy().astype(np.float64) * 2 ** 15, window, frame_length=400, hop_length=160, fft_length=512,
                             power=2.0, center=False, preemphasis=0.97, mel_filters=mel, log_mel="log",
                             mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
        assert mine.shape == ref.shape == (1 + (n - 400) // 160, 128)
        d = np.abs(mine - ref)
        assert float(d.max()) < 5e-3 and float(d.mean()) < 2e-5, (kind, n, float(d.max()), float(d.mean()))
