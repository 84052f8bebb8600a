"""UrbanSound8K front-end and dataset, mirroring /root/reference/.charles/spectrogram.py.

* module constants `SAMPLE_RATE, N_MELS, N_FFT, HOP_LENGTH, FMIN, FMAX, DURATION` read from the environment with
  the reference's defaults (spectrogram.py:56-62);
* `mel_spectrogram_log(waveform)`  = `torch.log(mel_spectrogram(waveform) + 1e-6)` (spectrogram.py:79-87,160-162),
  computed by libawt's `awt_logmel_generic` (periodic Hann, center / reflect, power 2, HTK mel, no norm);
* `prepare_waveform` = mono mean + `Resample(sr -> SAMPLE_RATE)` + pad / trim to DURATION (spectrogram.py:145-157);
  libawt's `awt_prepare_waveform` whenever the clip is on the GPU or needs resampling;
* `read_wav` + `preprocess_to_parquet` = the preprocessing loop (spectrogram.py:120-182): WAV samples go to the GPU as
  they lie in the file, one `awt_prepare_waveform` launch per file, one `awt_logmel_generic` launch per batch of files;
* `UrbanSoundDataSet(parquet_path=None, folds=None)` with `.df`, `.n_mels`, `__len__`, `__getitem__ ->
  (FloatTensor[n_mels, T], int)` over the reference's Parquet schema `rel_path, fold, class_id, class_name,
  log_mel_flat, log_mel_shape` (spectrogram.py:166-173,184-212).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
from torch.utils.data import Dataset

from . import _lib

SAMPLE_RATE = int(os.getenv("SAMPLE_RATE", 16000))
N_MELS = int(os.getenv("N_MELS", 128))
N_FFT = int(os.getenv("N_FFT", 1024))
HOP_LENGTH = int(os.getenv("HOP_LENGTH", 512))
FMIN = int(os.getenv("FMIN", 0))
FMAX = int(os.getenv("FMAX", 8000))
DURATION = float(os.getenv("DURATION", 4.0))
PROCESSED_PARQUET_PATH = os.getenv("PROCESSED_PARQUET_PATH", "./.data/UrbanSound8K/processed")
DATA_ROOT = os.getenv("DATA_ROOT", "./.data/UrbanSound8K")
METADATA_CSV = os.getenv("METADATA_CSV", "./.data/UrbanSound8K/metadata/UrbanSound8K.csv")


def get_processed_parquet_filename(n_mels: int = None, hop: int = None) -> str:
    """spectrogram.py:94-96."""
    return f"urbansound8k_processed_mels{n_mels or N_MELS}_hop{hop or HOP_LENGTH}.parquet"


def get_processed_parquet_path() -> str:
    return os.path.join(PROCESSED_PARQUET_PATH, get_processed_parquet_filename())


_PREPARED: set = set()     # front-end configurations whose device tables exist (awt_logmel_prepare / awt_resample_prepare)


def prepare_waveform(waveform: torch.Tensor, sample_rate: int = SAMPLE_RATE, duration: float = DURATION,
                     target_rate: int = SAMPLE_RATE, interleaved: bool = False, n_out: Optional[int] = None) -> torch.Tensor:
    """[C, n] or [n] at `sample_rate` -> [1, int(target_rate * duration)] fp32: channel mean, resample to `target_rate`
    (torchaudio.transforms.Resample defaults), zero-pad or truncate (spectrogram.py:145-157, in that order).

    A CPU tensor already at the target rate takes the reference's three host tensor ops.  Anything else -- a device
    tensor (int16 or float32; `interleaved=True` for WAV-order [n, C] data) or a clip that needs resampling -- goes
    through libawt (`awt_prepare_waveform`) and comes back as a device tensor; there is no host resampler."""
    n_out = int(target_rate * duration) if n_out is None else int(n_out)
    w = waveform
    if not w.is_cuda and int(sample_rate) == int(target_rate) and not interleaved:
        w = w.float()
        if w.dim() == 1:
            w = w.unsqueeze(0)
        if w.shape[0] > 1:
            w = torch.mean(w, dim=0, keepdim=True)
        if w.shape[1] < n_out:
            w = torch.nn.functional.pad(w, (0, n_out - w.shape[1]))
        else:
            w = w[:, :n_out]
        return w
    if w.dtype != torch.int16:
        w = w.float()
    if w.dim() == 1:
        w = w.unsqueeze(1) if interleaved else w.unsqueeze(0)
    if not w.is_cuda:
        w = w.cuda()
    w = w.contiguous()
    if interleaved:
        n_in, channels = w.shape
        cstride, sstride = 1, channels
    else:
        channels, n_in = w.shape
        cstride, sstride = n_in, 1
    out = torch.empty((1, n_out), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        key = ("rs", w.device.index, int(sample_rate), int(target_rate))
        if key not in _PREPARED:       # polyphase taps of this rate pair: built once, outside the compute call
            _lib.check(_lib.lib().awt_resample_prepare(_lib.ctx(w.device), int(sample_rate), int(target_rate)))
            _PREPARED.add(key)
        _lib.check(_lib.lib().awt_prepare_waveform(_lib.ctx(w.device), _lib.ptr(w), int(w.dtype == torch.int16), channels, cstride,
                                                   sstride, n_in, int(sample_rate), int(target_rate), _lib.ptr(out), n_out,
                                                   _lib.stream_handle()))
    return out


def preprocess_audio_for_cnn(waveform: torch.Tensor, sr: int):
    """spectrogram.py:214-240: ([C, n] waveform, sr) -> (mono waveform at SAMPLE_RATE in full length, its DURATION-second
    pad / trim, SAMPLE_RATE)."""
    n_in = waveform.shape[-1]
    n_full = n_in if int(sr) == SAMPLE_RATE else resampled_length(n_in, sr)
    full = prepare_waveform(waveform, sample_rate=sr, n_out=n_full)
    n = int(SAMPLE_RATE * DURATION)
    cnn = torch.nn.functional.pad(full, (0, n - n_full)) if n_full < n else full[:, :n].clone()
    return full, cnn, SAMPLE_RATE


def resampled_length(n_in: int, sample_rate: int, target_rate: int = SAMPLE_RATE) -> int:
    """ceil(n_in * target_rate / sample_rate): the length torchaudio's Resample returns."""
    return int(_lib.lib().awt_resampled_length(int(n_in), int(sample_rate), int(target_rate)))


def mel_spectrogram_log(waveform: torch.Tensor, sample_rate: int = None, n_fft: int = None, hop_length: int = None,
                        n_mels: int = None, f_min: float = None, f_max: float = None, log_eps: float = 1e-6) -> torch.Tensor:
    """Device waveform [B, n] (or [n]) fp32 -> device [B, n_mels, 1 + n // hop] = ln(mel_power + 1e-6)."""
    sample_rate = sample_rate or SAMPLE_RATE
    n_fft, hop_length, n_mels = n_fft or N_FFT, hop_length or HOP_LENGTH, n_mels or N_MELS
    f_min = FMIN if f_min is None else f_min
    f_max = FMAX if f_max is None else f_max
    w = waveform.float()
    squeeze = w.dim() == 1
    if squeeze:
        w = w.unsqueeze(0)
    if not w.is_cuda:
        w = w.cuda()
    w = w.contiguous()
    B, n = w.shape
    out = torch.empty((B, n_mels, 1 + n // hop_length), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        key = ("mel", w.device.index, n_fft, n_mels, float(f_min), float(f_max), sample_rate)
        if key not in _PREPARED:       # device tables of this front-end configuration: built once, outside the compute call
            _lib.check(_lib.lib().awt_logmel_prepare(_lib.ctx(w.device), n_fft, n_mels, float(f_min), float(f_max), sample_rate, 0))
            _PREPARED.add(key)
        _lib.check(_lib.lib().awt_logmel_generic(_lib.ctx(w.device), _lib.ptr(w), w.stride(0), B, n, sample_rate, n_fft, hop_length,
                                                 n_mels, float(f_min), float(f_max), float(log_eps), _lib.ptr(out),
                                                 _lib.stream_handle()))
    return out[0] if squeeze else out


def record_for_parquet(rel_path: str, fold: int, class_id: int, class_name: str, log_mel: torch.Tensor) -> dict:
    """One row of the reference's Parquet schema (spectrogram.py:165-173)."""
    a = log_mel.detach().cpu().numpy().astype(np.float32)
    return {"rel_path": rel_path, "fold": fold, "class_id": class_id, "class_name": class_name,
            "log_mel_flat": a.flatten(), "log_mel_shape": list(a.shape)}


def read_wav(path: str):
    """RIFF/WAVE reader for what UrbanSound8K ships as plain samples: integer PCM of 8 / 16 / 24 / 32 bits and IEEE float
    32 / 64 (format tags 1, 3 and their WAVE_FORMAT_EXTENSIBLE forms).  Returns (samples [n, C] in file order, rate):
    int16 for 16-bit PCM (the GPU scales it by 1/32768 like torchaudio.load), float32 in [-1, 1) otherwise.
    Compressed formats (ADPCM, ...) raise ValueError; the preprocessing loop logs and skips such files as the
    reference does with any loader error (spectrogram.py:174-175)."""
    import struct

    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos: pos + 4], struct.unpack("<I", data[pos + 4: pos + 8])[0]
        body = data[pos + 8: pos + 8 + size]
        if tag == b"fmt ":
            fmt = body
        elif tag == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None or len(fmt) < 16:
        raise ValueError(f"{path}: missing fmt or data chunk")
    code, channels, rate, _, align, bits = struct.unpack("<HHIIHH", fmt[:16])
    if code == 0xFFFE and len(fmt) >= 26:                      # WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the real tag
        code = struct.unpack("<H", fmt[24:26])[0]
    if channels < 1 or align != channels * bits // 8:
        raise ValueError(f"{path}: inconsistent fmt chunk")
    n = len(pcm) // align
    raw = np.frombuffer(pcm, dtype=np.uint8, count=n * align)
    if code == 1 and bits == 16:
        x = raw.view("<i2").reshape(n, channels).copy()
    elif code == 1 and bits == 8:
        x = ((raw.astype(np.float32) - 128.0) / 128.0).reshape(n, channels)
    elif code == 1 and bits == 24:
        b = raw.reshape(n * channels, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = (v.astype(np.float32) / float(1 << 23)).reshape(n, channels)
    elif code == 1 and bits == 32:
        x = (raw.view("<i4").astype(np.float64) / float(1 << 31)).astype(np.float32).reshape(n, channels)
    elif code == 3 and bits in (32, 64):
        x = raw.view("<f4" if bits == 32 else "<f8").astype(np.float32).reshape(n, channels)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {code} with {bits} bits")
    return torch.from_numpy(x), int(rate)


def preprocess_to_parquet(metadata_csv: Optional[str] = None, data_root: Optional[str] = None, parquet_path: Optional[str] = None,
                          overwrite: bool = False, loader=None, batch_files: int = 64, device: str = "cuda") -> Optional[str]:
    """UrbanSound8K audio -> log-mel spectrograms -> Parquet, as spectrogram.py:120-182 (same columns, same row order).

    `loader(path) -> (samples, rate)` defaults to `read_wav` ([n, C] file order); a torchaudio-style loader returning
    [C, n] float tensors works too (`interleaved` is inferred from which axis is longer than 8).  Instead of prompting
    on an existing file the function returns None unless `overwrite=True`."""
    import pandas as pd
    import pyarrow.parquet  # noqa: F401  (the Parquet engine pandas uses; imported here so a broken install fails with its own message)

    parquet_path = parquet_path or get_processed_parquet_path()
    if os.path.exists(parquet_path) and not overwrite:
        return None
    data_root = data_root or DATA_ROOT
    df = pd.read_csv(metadata_csv or METADATA_CSV)
    loader = loader or read_wav
    n_out = int(SAMPLE_RATE * DURATION)
    records, pending, meta = [], [], []

    def flush():
        if not pending:
            return
        mels = mel_spectrogram_log(torch.cat(pending, dim=0))          # one launch for the whole batch of files
        for m, (rel_path, fold, class_id, class_name) in zip(mels, meta):
            records.append(record_for_parquet(rel_path, fold, class_id, class_name, m))
        pending.clear(); meta.clear()

    for _, row in df.iterrows():
        rel_path = os.path.join("audio", f"fold{row['fold']}", row["slice_file_name"])
        try:
            samples, rate = loader(os.path.join(data_root, rel_path))
            interleaved = samples.dim() == 2 and samples.shape[1] <= 8 < samples.shape[0]
            w = prepare_waveform(samples.to(device), sample_rate=rate, interleaved=interleaved)
            assert tuple(w.shape) == (1, n_out)
        except Exception as e:   # the reference logs and continues (spectrogram.py:174-175)
            print(f"[urbansound] error processing {rel_path}: {e}")
            continue
        pending.append(w)
        meta.append((rel_path, int(row["fold"]), int(row["classID"]), row["class"]))
        if len(pending) >= batch_files:
            flush()
    flush()
    out_df = pd.DataFrame(records)
    os.makedirs(os.path.dirname(os.path.abspath(parquet_path)), exist_ok=True)
    out_df.to_parquet(parquet_path, index=False)
    return parquet_path


class UrbanSoundDataSet(Dataset):
    def __init__(self, parquet_path: Optional[str] = None, folds=None):
        import pandas as pd
        import pyarrow.parquet  # noqa: F401  (see preprocess_to_parquet)

        if parquet_path is None:
            parquet_path = get_processed_parquet_path()
        self.df = pd.read_parquet(parquet_path)
        if folds is not None:
            self.df = self.df[self.df["fold"].isin(folds)].reset_index(drop=True)
        self.n_mels = N_MELS

    def __len__(self):
        return len(self.df)

    def __getitem__(self, idx):
        row = self.df.iloc[idx]
        log_mel = np.array(row["log_mel_flat"], dtype=np.float32).reshape(tuple(row["log_mel_shape"]))
        return torch.tensor(log_mel), int(row["class_id"])
