"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's log-mel front-ends.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker / reported CPU baseline -- never as the thing measured or shipped.  The product
path (mlx8-ws-audio-transformer_amd/) never imports it and fails loudly without its HIP library.

What the reference computes (it has no arithmetic of its own -- SURVEY.md §0.3):

* Whisper log-mel: `processor(audio, sampling_rate=16000)` at /root/reference/AB/fineTune.py:88,
  AB/wavToWhisper.py:55, AB/fineTuneMidiTester.py:33, .charles/music2midi/model.py:100-104
  -> transformers `WhisperFeatureExtractor.__call__` (HF:models/whisper/feature_extraction_whisper.py:193-346)
  -> `_np_extract_fbank_features` (:105-133) / `_torch_extract_fbank_features` (:135-168)
  -> `audio_utils.spectrogram` (HF:audio_utils.py:809-1017) and `mel_filter_bank` (:636-729).
  transformers is a pinned third-party dependency of the reference (AB/pyproject.toml:22 `==4.35.2`,
  .charles/uv.lock 4.53.1; 5.15.0 is what this image has).  Its published algorithm is restated here
  in NumPy float64 (the only path 4.35.2 had), and pinned against the installed package's outputs by
  tools/make_golden.py -> tests/golden/logmel_*.npz (parity pinned to transformers 5.15.0).
* UrbanSound log-mel: /root/reference/.charles/spectrogram.py:79-87 (torchaudio MelSpectrogram,
  power 2, HTK mel, no norm) + :145-162 (mono/pad/trim to 64000, `log(mel + 1e-6)`).  torchaudio is
  not installed in this image, so this front-end is "parity unpinned" against torchaudio itself;
  it is pinned to the same triangular-filter construction (HF mel_filter_bank htk / norm=None) and
  to torch.stft on the CPU (SURVEY.md §8c row "does not pin" (2)).
"""
from __future__ import annotations

import numpy as np

WHISPER_SR = 16000
WHISPER_N_FFT = 400
WHISPER_HOP = 160
WHISPER_N_MELS = 80
WHISPER_N_SAMPLES = 480000  # chunk_length 30 s * 16 kHz (HF:feature_extraction_whisper.py:91)


# ----------------------------------------------------------------------------- mel scales
def hertz_to_mel(freq, mel_scale: str = "htk"):
    """HF:audio_utils.py:448-481."""
    freq = np.asarray(freq, dtype=np.float64)
    if mel_scale == "htk":
        return 2595.0 * np.log10(1.0 + freq / 700.0)
    if mel_scale != "slaney":
        raise ValueError("mel_scale must be 'htk' or 'slaney'")
    min_log_hertz, min_log_mel = 1000.0, 15.0
    logstep = 27.0 / np.log(6.4)
    mels = 3.0 * freq / 200.0
    log_region = freq >= min_log_hertz
    safe = np.where(log_region, freq, min_log_hertz)
    return np.where(log_region, min_log_mel + np.log(safe / min_log_hertz) * logstep, mels)


def mel_to_hertz(mels, mel_scale: str = "htk"):
    """HF:audio_utils.py:484-517."""
    mels = np.asarray(mels, dtype=np.float64)
    if mel_scale == "htk":
        return 700.0 * (np.power(10, mels / 2595.0) - 1.0)
    if mel_scale != "slaney":
        raise ValueError("mel_scale must be 'htk' or 'slaney'")
    min_log_hertz, min_log_mel = 1000.0, 15.0
    logstep = np.log(6.4) / 27.0
    freq = 200.0 * mels / 3.0
    log_region = mels >= min_log_mel
    return np.where(log_region, min_log_hertz * np.exp(logstep * (mels - min_log_mel)), freq)


def mel_filter_bank(num_frequency_bins: int, num_mel_filters: int, min_frequency: float, max_frequency: float,
                    sampling_rate: int, norm: str | None = None, mel_scale: str = "htk") -> np.ndarray:
    """float64 [num_frequency_bins, num_mel_filters] triangular bank (HF:audio_utils.py:636-729, 541-559)."""
    mel_min = hertz_to_mel(min_frequency, mel_scale)
    mel_max = hertz_to_mel(max_frequency, mel_scale)
    mel_freqs = np.linspace(mel_min, mel_max, num_mel_filters + 2)
    filter_freqs = mel_to_hertz(mel_freqs, mel_scale)
    fft_freqs = np.linspace(0, sampling_rate // 2, num_frequency_bins)
    filter_diff = np.diff(filter_freqs)
    slopes = np.expand_dims(filter_freqs, 0) - np.expand_dims(fft_freqs, 1)
    down = -slopes[:, :-2] / filter_diff[:-1]
    up = slopes[:, 2:] / filter_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    if norm == "slaney":
        enorm = 2.0 / (filter_freqs[2: num_mel_filters + 2] - filter_freqs[:num_mel_filters])
        fb = fb * np.expand_dims(enorm, 0)
    elif norm is not None:
        raise ValueError("norm must be None or 'slaney'")
    return fb


def whisper_mel_filters(n_mels: int = WHISPER_N_MELS) -> np.ndarray:
    """HF:feature_extraction_whisper.py:95-103 (`feature_size` = 80, or 128 for large-v3)."""
    return mel_filter_bank(1 + WHISPER_N_FFT // 2, n_mels, 0.0, 8000.0, WHISPER_SR, "slaney", "slaney")


def hann_periodic(n: int) -> np.ndarray:
    """`window_function(n, "hann")` = np.hanning(n + 1)[:-1] (HF:audio_utils.py:778-792) == torch.hann_window(n)."""
    return np.hanning(n + 1)[:-1]


# ----------------------------------------------------------------------------- STFT power
def stft_power(waveform: np.ndarray, n_fft: int, hop: int, dtype=np.float64) -> np.ndarray:
    """|STFT|^2 with center=True / reflect padding, periodic Hann, onesided.  -> [n_fft//2+1, 1 + len//hop].

    float64 follows `audio_utils.spectrogram` (HF:audio_utils.py:940-985: the per-frame rfft result is
    stored as complex64, then |.|^2 in float64); float32 follows `torch.stft` in fp32
    (HF:feature_extraction_whisper.py:149-154)."""
    x = np.pad(np.asarray(waveform, dtype=np.float64), (n_fft // 2, n_fft // 2), mode="reflect")
    n_frames = 1 + (x.size - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = x[idx] * hann_periodic(n_fft)[None, :]
    if dtype == np.float32:
        spec = np.fft.rfft(frames.astype(np.float32), axis=1).astype(np.complex64)
        return (np.abs(spec).astype(np.float32) ** 2).T
    spec = np.fft.rfft(frames, axis=1).astype(np.complex64)
    return (np.abs(spec, dtype=np.float64) ** 2).T


# ----------------------------------------------------------------------------- Whisper front-end
def pad_or_trim(waveform: np.ndarray, n_samples: int) -> np.ndarray:
    """Zero-pad on the right / truncate to `n_samples` (HF:feature_extraction_whisper.py:300-307)."""
    w = np.asarray(waveform, dtype=np.float32).reshape(-1)
    if w.size >= n_samples:
        return w[:n_samples]
    return np.concatenate([w, np.zeros(n_samples - w.size, dtype=np.float32)])


def whisper_logmel(clips, n_samples: int = WHISPER_N_SAMPLES, n_mels: int = WHISPER_N_MELS) -> np.ndarray:
    """[B, n_mels, n_samples // 160] fp32 Whisper input features for a list of mono fp32 clips.

    n_samples = 480000 is the reference's behaviour (parity mode, T = 3000);
    n_samples = 64000 is the trimmed mode (T = 400) which the reference never computes.
    Follows `_np_extract_fbank_features` (HF:feature_extraction_whisper.py:105-133).
    """
    if isinstance(clips, np.ndarray) and clips.ndim == 1:
        clips = [clips]
    filters = whisper_mel_filters(n_mels)
    out = []
    for clip in clips:
        w = pad_or_trim(clip, n_samples)
        power = stft_power(w, WHISPER_N_FFT, WHISPER_HOP)                  # [201, T+1]
        mel = np.maximum(1e-10, filters.T @ power)                          # mel_floor (HF:audio_utils.py:990)
        log_spec = np.log10(mel).astype(np.float32)[:, :-1]                 # dtype cast then drop last frame
        log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
        log_spec = (log_spec + 4.0) / 4.0
        out.append(log_spec)
    return np.asarray(out, dtype=np.float32)


# ----------------------------------------------------------------------------- UrbanSound front-end
def urbansound_prepare(waveform: np.ndarray, sample_rate: int = 16000, duration: float = 4.0) -> np.ndarray:
    """Channel mean + zero-pad / truncate to int(sr * duration) (/root/reference/.charles/spectrogram.py:145-157).

    `waveform` is [C, n] or [n] at `sample_rate` already (resampling is out of scope)."""
    w = np.asarray(waveform, dtype=np.float32)
    if w.ndim == 2:
        w = w.mean(axis=0) if w.shape[0] > 1 else w[0]
    return pad_or_trim(w, int(sample_rate * duration))


def urbansound_logmel(waveform: np.ndarray, sample_rate: int = 16000, n_fft: int = 1024, hop: int = 512,
                      n_mels: int = 128, f_min: float = 0.0, f_max: float = 8000.0, log_eps: float = 1e-6) -> np.ndarray:
    """[n_mels, 1 + n // hop] fp32 = ln(mel_power + 1e-6)  (/root/reference/.charles/spectrogram.py:79-87,160-162).

    torchaudio.transforms.MelSpectrogram defaults: win_length = n_fft, periodic Hann, center, reflect,
    power 2, HTK scale, norm None, fb built on linspace(0, sr//2, n_fft//2+1)."""
    fb = mel_filter_bank(n_fft // 2 + 1, n_mels, f_min, f_max, sample_rate, None, "htk")
    power = stft_power(np.asarray(waveform, dtype=np.float32).reshape(-1), n_fft, hop)
    mel = fb.T @ power
    return np.log(mel + log_eps).astype(np.float32)
