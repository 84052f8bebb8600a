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


def test_preprocess_audio_for_cnn_returns_full_and_fixed_length():
    x = _clip(22050, 5.5, 2, seed=9)
    full, cnn, sr = urbansound.preprocess_audio_for_cnn(torch.from_numpy(x), 22050)
    ref = R.resample(x.mean(axis=0, dtype=np.float32), 22050, 16000)
    assert sr == 16000 and tuple(full.shape) == (1, len(ref)) and tuple(cnn.shape) == (1, 64000)
    np.testing.assert_allclose(full[0].cpu().numpy(), ref, rtol=0, atol=TOL)
    np.testing.assert_array_equal(cnn.cpu().numpy(), full[:, :64000].cpu().numpy())
    full, cnn, _ = urbansound.preprocess_audio_for_cnn(torch.from_numpy(_clip(16000, 1.0, 1, seed=10)), 16000)
    assert tuple(full.shape) == (1, 16000) and tuple(cnn.shape) == (1, 64000) and not cnn[:, 16000:].any()


def test_bad_arguments_raise():
    from mlx8_ws_audio_transformer_amd import _lib
    with pytest.raises(_lib.AwtError):
        urbansound.prepare_waveform(torch.zeros(1, 100), sample_rate=500)
    with pytest.raises(_lib.AwtError):
        urbansound.prepare_waveform(torch.zeros(9, 100), sample_rate=44100)


def test_preprocess_to_parquet_roundtrip(tmp_path):
    """spectrogram.py:120-182 end to end: WAV files of mixed rate / width / channels -> Parquet -> UrbanSoundDataSet."""
    import struct
    import wave

    import pandas as pd

    root = tmp_path / "UrbanSound8K"
    (root / "audio" / "fold1").mkdir(parents=True)
    (root / "audio" / "fold2").mkdir(parents=True)
    clips = {"a.wav": (44100, _clip(44100, 2.5, 2, 1)), "b.wav": (8000, _clip(8000, 1.0, 1, 2)), "c.wav": (16000, _clip(16000, 5.0, 1, 3))}
    refs = {}
    for name, (sr, x) in clips.items():
        i16 = np.round(x * 20000).astype(np.int16)
        with wave.open(str(root / "audio" / ("fold2" if name == "c.wav" else "fold1") / name), "wb") as w:
            w.setnchannels(i16.shape[0]); w.setsampwidth(2); w.setframerate(sr)
            w.writeframes(np.ascontiguousarray(i16.T).tobytes())
        refs[name] = oracle_logmel.urbansound_logmel(R.prepare_waveform(i16.astype(np.float32) / 32768.0, sr), n_mels=urbansound.N_MELS,
                                                     hop=urbansound.HOP_LENGTH)
    (root / "audio" / "fold1" / "broken.wav").write_bytes(b"RIFF....WAVEjunk")
    meta = pd.DataFrame({"slice_file_name": ["a.wav", "broken.wav", "b.wav", "c.wav"], "fold": [1, 1, 1, 2], "classID": [3, 0, 7, 9],
                         "class": ["dog_bark", "x", "jackhammer", "street_music"]})
    meta.to_csv(tmp_path / "meta.csv", index=False)
    out = urbansound.preprocess_to_parquet(str(tmp_path / "meta.csv"), str(root), str(tmp_path / "out" / "p.parquet"), batch_files=2)
    assert out and urbansound.preprocess_to_parquet(str(tmp_path / "meta.csv"), str(root), out) is None      # exists: not overwritten
    ds = urbansound.UrbanSoundDataSet(out)
    assert len(ds) == 3 and list(ds.df["class_id"]) == [3, 7, 9]                                            # the broken file is skipped
    assert list(ds.df.columns) == ["rel_path", "fold", "class_id", "class_name", "log_mel_flat", "log_mel_shape"]
    for i, name in enumerate(["a.wav", "b.wav", "c.wav"]):
        mel, label = ds[i]
        assert tuple(mel.shape) == refs[name].shape
        # b.wav is UP-sampled: its mel bins above 4 kHz hold only the resampler's stop-band residue (power ~1e-9 next to the
        # 1e-6 inside the log), where fp32-vs-float64 rounding of the filter sum moves ln(p + 1e-6) by ~1e-3
        err = np.abs(mel.numpy() - refs[name]).max()
        assert err < (5e-3 if name == "b.wav" else 2e-4), (name, err)
    assert len(urbansound.UrbanSoundDataSet(out, folds=[2])) == 1
