"""awt_prepare_waveform (mono mix + torchaudio-style resampling + pad / trim on the GPU) against oracle/resample.py."""
import math

import numpy as np
import pytest
import torch

from oracle import logmel as oracle_logmel, resample as R
from mlx8_ws_audio_transformer_amd import urbansound

pytestmark = pytest.mark.gpu
TOL = 2e-6     # fp32 accumulation of <= 73 taps of |x| <= 1 data vs the float64 oracle


def _clip(sr, seconds, channels, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(int(sr * seconds)) / sr
    x = np.stack([0.4 * np.sin(2 * np.pi * (220.0 * (c + 1)) * t) + 0.1 * rng.standard_normal(t.shape) for c in range(channels)])
    return x.astype(np.float32)


@pytest.mark.parametrize("sr", [8000, 11025, 22050, 32000, 44100, 48000, 96000])
@pytest.mark.parametrize("channels", [1, 2])
def test_resample_matches_oracle(sr, channels):
    x = _clip(sr, 1.3, channels, seed=sr + channels)
    got = urbansound.prepare_waveform(torch.from_numpy(x), sample_rate=sr, duration=4.0)
    assert got.is_cuda and tuple(got.shape) == (1, 64000)
    ref = R.prepare_waveform(x, sr, 16000, 4.0)
    assert urbansound.resampled_length(x.shape[1], sr) == math.ceil(x.shape[1] * 16000 / sr)
    np.testing.assert_allclose(got[0].cpu().numpy(), ref, rtol=0, atol=TOL)


def test_truncation_int16_and_interleaved():
    sr = 44100
    x = _clip(sr, 6.0, 2, seed=7)                       # longer than 4 s: truncated after resampling
    i16 = np.round(x * 20000).astype(np.int16)
    ref = R.prepare_waveform(i16.astype(np.float32) / 32768.0, sr, 16000, 4.0)
    planar = urbansound.prepare_waveform(torch.from_numpy(i16), sample_rate=sr)
    inter = urbansound.prepare_waveform(torch.from_numpy(np.ascontiguousarray(i16.T)), sample_rate=sr, interleaved=True)
    np.testing.assert_allclose(planar[0].cpu().numpy(), ref, rtol=0, atol=TOL)
    np.testing.assert_array_equal(planar.cpu().numpy(), inter.cpu().numpy())


def test_equal_rate_device_path_is_exact_and_feeds_logmel():
    x = _clip(16000, 2.0, 2, seed=3)
    got = urbansound.prepare_waveform(torch.from_numpy(x).cuda())
    ref = oracle_logmel.urbansound_prepare(x)
    np.testing.assert_array_equal(got[0].cpu().numpy(), ref)
    # file -> resample -> log-mel entirely on the device (spectrogram.py:145-162)
    x48 = _clip(48000, 4.0, 2, seed=4)
    w = urbansound.prepare_waveform(torch.from_numpy(x48), sample_rate=48000)
    mel = urbansound.mel_spectrogram_log(w, n_mels=80).cpu().numpy()[0]
    ref = oracle_logmel.urbansound_logmel(R.prepare_waveform(x48, 48000), n_mels=80)
    np.testing.assert_allclose(mel, ref, rtol=0, atol=2e-4)   # resampler fp32 rounding (2e-6) through ln(power + 1e-6)


def test_bad_arguments_raise():
    from mlx8_ws_audio_transformer_amd import _lib
    with pytest.raises(_lib.AwtError):
        urbansound.prepare_waveform(torch.zeros(1, 100), sample_rate=500)
    with pytest.raises(_lib.AwtError):
        urbansound.prepare_waveform(torch.zeros(9, 100), sample_rate=44100)
