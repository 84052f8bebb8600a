"""The resampling oracle (oracle/resample.py) restates torchaudio's sinc_interp_hann resampler, which is not installed
here ("parity unpinned" against torchaudio itself).  It is pinned by what the algorithm guarantees and by an independent
evaluation of the same filter through torch.nn.functional.conv1d (the operation torchaudio itself uses)."""
import math

import numpy as np
import pytest
import torch

from oracle import resample as R

RATES = [8000, 11025, 22050, 32000, 44100, 48000, 96000]


@pytest.mark.parametrize("sr", RATES)
def test_kernel_shape_and_gain(sr):
    k, width, orig, new = R.sinc_resample_kernel(sr, 16000)
    g = math.gcd(sr, 16000)
    assert (orig, new) == (sr // g, 16000 // g)
    assert width == math.ceil(6 * orig / (min(orig, new) * 0.99))
    assert k.shape == (new, 2 * width + orig) and k.dtype == np.float32
    np.testing.assert_allclose(k.sum(axis=1), 1.0, atol=1e-3)      # unit DC gain of every phase (Hann-windowed sinc: +4.7e-4)
    # taps outside the window are exactly zero: |t| >= 6 after the clamp gives cos(pi/2)^2 -> < 1e-32
    t = ((-np.arange(new)[:, None] / new) + np.arange(-width, width + orig)[None, :] / orig) * min(orig, new) * 0.99
    assert np.all(np.abs(k[np.abs(t) >= 6.0]) < 1e-30)


@pytest.mark.parametrize("sr", RATES)
def test_matches_conv1d_formulation(sr):
    rng = np.random.default_rng(sr)
    x = rng.standard_normal(5000).astype(np.float32)
    k, width, orig, new = R.sinc_resample_kernel(sr, 16000)
    xp = torch.nn.functional.pad(torch.from_numpy(x)[None, None], (width, width + orig))
    y = torch.nn.functional.conv1d(xp, torch.from_numpy(k)[:, None], stride=orig)           # [1, new, steps]
    y = y.transpose(1, 2).reshape(-1)[: math.ceil(new * len(x) / orig)].numpy()
    got = R.resample(x, sr, 16000)
    assert got.shape == y.shape
    np.testing.assert_allclose(got, y, rtol=0, atol=5e-6)


@pytest.mark.parametrize("sr", [8000, 22050, 44100, 48000])
def test_band_limited_sine_survives(sr):
    n = sr  # 1 s
    f = 1234.5
    x = (0.5 * np.sin(2 * np.pi * f * np.arange(n) / sr)).astype(np.float32)
    y = R.resample(x, sr, 16000)
    assert len(y) == math.ceil(n * 16000 / sr)
    ref = 0.5 * np.sin(2 * np.pi * f * np.arange(len(y)) / 16000)
    np.testing.assert_allclose(y[300:-300], ref[300:-300], atol=2e-3)   # windowed-sinc passband ripple (largest when up-sampling)


def test_equal_rates_is_identity_and_prepare_rules():
    x = np.random.default_rng(0).standard_normal((2, 1000)).astype(np.float32)
    np.testing.assert_array_equal(R.resample(x, 16000, 16000), x)
    w = R.prepare_waveform(x, 16000, 16000, duration=0.1)
    assert w.shape == (1600,)
    np.testing.assert_array_equal(w[:1000], x.mean(axis=0, dtype=np.float32))
    assert not w[1000:].any()
    # down-sampling a long stereo clip: mean first, then resample, then truncate
    x = np.random.default_rng(1).standard_normal((2, 44100)).astype(np.float32)
    w = R.prepare_waveform(x, 44100, 16000, duration=0.5)
    np.testing.assert_array_equal(w, R.resample(x.mean(axis=0, dtype=np.float32), 44100, 16000)[:8000])
