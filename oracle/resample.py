"""CPU restatement of the mono-mix + sample-rate conversion + pad/trim step of the UrbanSound front-end.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
path (mlx8_ws_audio_transformer_amd) never imports this package.

Reference call site: /root/reference/.charles/spectrogram.py:145-157
    waveform, sr = torchaudio.load(path)
    if MONO and waveform.shape[0] > 1: waveform = torch.mean(waveform, dim=0, keepdim=True)
    if sr != SAMPLE_RATE: waveform = torchaudio.transforms.Resample(orig_freq=sr, new_freq=SAMPLE_RATE)(waveform)
    pad with zeros / truncate to int(SAMPLE_RATE * DURATION)

The arithmetic lives in torchaudio (locked 2.7.1 in /root/reference/.charles/uv.lock, NOT installed in this image, not
vendored): `torchaudio.functional._get_sinc_resample_kernel` / `_apply_sinc_resample_kernel` with the transform's
defaults (resampling_method "sinc_interp_hann", lowpass_filter_width 6, rolloff 0.99, dtype None).  **Parity with
torchaudio itself is unpinned** (SURVEY.md §8c item 2): this file restates the published algorithm --

    orig, new = sr_in / gcd, sr_out / gcd;  base = min(orig, new) * rolloff;  width = ceil(lpw * orig / base)
    idx = arange(-width, width + orig) / orig                     (float64)
    t   = (arange(0, -new, -1)[:, None] / new + idx) * base       (phase term divided in float32: an int64 tensor / int)
    t   = clamp(t, -lpw, lpw);  window = cos(t pi / lpw / 2)^2;  kernel = sinc(t pi) * window * base / orig  -> float32
    y   = conv1d(pad(x, (width, width + orig)), kernel[new, 1, K], stride = orig), interleaved, cut to ceil(new n / orig)

-- and is pinned by the properties the algorithm guarantees (tests/test_oracle_resample.py): identity kernel for equal
rates, unit DC gain, a band-limited sine comes out as the same sine at the new rate, output length formula.
"""
from __future__ import annotations

import math

import numpy as np

LOWPASS_FILTER_WIDTH = 6
ROLLOFF = 0.99


def sinc_resample_kernel(sr_in: int, sr_out: int, lowpass_filter_width: int = LOWPASS_FILTER_WIDTH,
                         rolloff: float = ROLLOFF):
    """-> (kernel float32 [new, K], width, orig, new) with K = 2 * width + orig."""
    g = math.gcd(int(sr_in), int(sr_out))
    orig, new = int(sr_in) // g, int(sr_out) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    phase = (np.arange(0, -new, -1).astype(np.float32) / np.float32(new)).astype(np.float64)[:, None]
    t = (phase + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    k = k * window * (base / orig)
    return k.astype(np.float32), width, orig, new


def resample(x: np.ndarray, sr_in: int, sr_out: int) -> np.ndarray:
    """[..., n] float -> [..., ceil(n * new / orig)] float32 (accumulated in float64, rounded once)."""
    x = np.asarray(x, dtype=np.float32)
    if int(sr_in) == int(sr_out):
        return x.copy()
    k, width, orig, new = sinc_resample_kernel(sr_in, sr_out)
    lead = x.shape[:-1]
    n = x.shape[-1]
    xp = np.pad(x.reshape(-1, n), ((0, 0), (width, width + orig))).astype(np.float64)
    K = k.shape[1]
    steps = (xp.shape[1] - K) // orig + 1
    frames = np.lib.stride_tricks.sliding_window_view(xp, K, axis=1)[:, ::orig][:, :steps]   # [W, steps, K]
    y = frames @ k.astype(np.float64).T                                                       # [W, steps, new]
    y = y.reshape(y.shape[0], -1)[:, : math.ceil(new * n / orig)]
    return y.reshape(*lead, -1).astype(np.float32)


def prepare_waveform(waveform: np.ndarray, sr_in: int, sr_out: int = 16000, duration: float = 4.0) -> np.ndarray:
    """[C, n] or [n] at sr_in -> [int(sr_out * duration)] float32: channel mean, resample, zero-pad / truncate
    (/root/reference/.charles/spectrogram.py:146-157, in that order)."""
    w = np.asarray(waveform, dtype=np.float32)
    if w.ndim == 2:
        w = w.mean(axis=0, dtype=np.float32) if w.shape[0] > 1 else w[0]
    w = resample(w, sr_in, sr_out)
    n = int(sr_out * duration)
    out = np.zeros(n, dtype=np.float32)
    out[: min(n, w.shape[0])] = w[:n]
    return out
