"""Host-side mirror of the reference's audio front-end call surface, backed by the HIP log-mel kernels.

Reference interface kept (names, argument meaning, error behaviour):

* `processor(audio_array, sampling_rate=16000[, text=...][, return_tensors="pt"])` -> mapping with
  `["input_features"]` of shape [1, 80, 3000] fp32 (and `["labels"]` when `text` is given)
      /root/reference/AB/fineTune.py:88, AB/wavToWhisper.py:55, AB/fineTuneMidiTester.py:33,
      /root/reference/.charles/music2midi/model.py:100-104 (reads `.input_features` as an attribute)
* `processor.feature_extractor.pad(list_of_{"input_features"}, return_tensors="pt")`   AB/fineTune.py:107
* `processor.tokenizer.pad(...)`, `processor.batch_decode(...)`                         AB/fineTune.py:110, AB/wavToWhisper.py:62
  -- the tokenizer is text plumbing outside the hot path: pass any HF tokenizer in, or leave it None.

Behaviour restated from transformers' `WhisperFeatureExtractor.__call__`
(HF:models/whisper/feature_extraction_whisper.py:193-346): ValueError unless sampling_rate == 16000 (:265-271),
mono only (:279-280), zero-pad / truncate to 30 s (:300-307), results returned on the CPU.
The arithmetic runs in libawt (`awt_logmel_whisper`); without the library or a GPU this module raises.
"""
from __future__ import annotations

from typing import Any, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .collator import stack_input_features


class BatchFeature(dict):
    """dict with attribute access, like transformers.BatchFeature (model.py:105 uses `inputs.input_features`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, *a, **kw):
        return BatchFeature({k: (v.to(*a, **kw) if isinstance(v, torch.Tensor) else v) for k, v in self.items()})


def logmel_whisper_device(pcm: torch.Tensor, n_valid: Optional[torch.Tensor] = None, max_valid: Optional[int] = None,
                          n_frames: int = 3000, out: Optional[torch.Tensor] = None, n_mels: int = 80) -> torch.Tensor:
    """Device PCM [B, n] (int16 or float32, contiguous rows) -> device float32 [B, n_mels, n_frames] (n_mels 80, or 128
    for large-v3).  No host round trip."""
    if pcm.dim() != 2 or pcm.dtype not in (torch.int16, torch.float32):
        raise ValueError("pcm must be a [B, n] int16 or float32 device tensor")
    B, n = pcm.shape
    if max_valid is None:
        max_valid = n
    if max_valid > n:
        raise ValueError("max_valid exceeds the clip stride")
    if n_valid is not None:
        n_valid = n_valid.to(device=pcm.device, dtype=torch.int32).contiguous()
        if n_valid.numel() != B:
            raise ValueError("n_valid must have one entry per clip")
    if out is None:
        out = torch.empty((B, n_mels, n_frames), dtype=torch.float32, device=pcm.device)
    L = _lib.lib()
    ws = _lib.workspace(L.awt_logmel_workspace_bytes(B), pcm.device)
    with torch.cuda.device(pcm.device):
        _lib.check(L.awt_logmel_whisper_mels(_lib.ctx(pcm.device), _lib.ptr(pcm), int(pcm.dtype == torch.int16), pcm.stride(0),
                                             _lib.ptr(n_valid), int(max_valid), B, n_frames, int(n_mels), _lib.ptr(out), _lib.ptr(ws),
                                             ws.numel(), _lib.stream_handle()))
    return out


class WhisperFeatureExtractor:
    """Whisper log-mel extractor with transformers' defaults (feature_extraction_whisper.py:69-103)."""

    model_input_names = ["input_features"]

    def __init__(self, feature_size: int = 80, sampling_rate: int = 16000, hop_length: int = 160, chunk_length: int = 30,
                 n_fft: int = 400, padding_value: float = 0.0, return_attention_mask: bool = False, device: str = "cuda"):
        if feature_size not in (80, 128) or (sampling_rate, hop_length, n_fft) != (16000, 160, 400):
            raise ValueError("the native extractor implements Whisper's front-end: 80 or 128 mels, 16 kHz, hop 160, n_fft 400")
        self.feature_size, self.sampling_rate, self.hop_length, self.n_fft = feature_size, sampling_rate, hop_length, n_fft
        self.chunk_length = chunk_length
        self.n_samples = chunk_length * sampling_rate
        self.nb_max_frames = self.n_samples // hop_length
        self.padding_value = padding_value
        self.return_attention_mask = return_attention_mask
        self.device = device

    # -- reference surface -------------------------------------------------------------------------------------
    def __call__(self, raw_speech, truncation: bool = True, pad_to_multiple_of=None, return_tensors: Optional[str] = None,
                 return_attention_mask: Optional[bool] = None, padding: Optional[str] = "max_length",
                 max_length: Optional[int] = None, sampling_rate: Optional[int] = None, do_normalize: Optional[bool] = None,
                 device: Optional[str] = None, keep_on_device: bool = False, **kwargs) -> BatchFeature:
        """`keep_on_device=True` (with return_tensors="pt") leaves `input_features` in HBM instead of copying it to the host as the
        reference's extractor does: the 960 KB per clip then never cross PCIe on their way to the encoder."""
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(
                f"The model corresponding to this feature extractor: {self.__class__.__name__} was trained using a"
                f" sampling rate of {self.sampling_rate}. Please make sure that the provided `raw_speech` input"
                f" was sampled with {self.sampling_rate} and not {sampling_rate}.")
        if do_normalize:
            raise NotImplementedError("do_normalize is not used by the reference and not implemented natively")
        if padding not in ("max_length", None) or pad_to_multiple_of is not None or not truncation:
            raise NotImplementedError("only padding='max_length' with truncation (the reference's call) is implemented")
        clips = self._to_clip_list(raw_speech)
        n_samples = int(max_length) if max_length else self.n_samples
        if n_samples % (4 * self.hop_length) != 0:
            raise ValueError("max_length must be a multiple of 640 samples")
        n_frames = n_samples // self.hop_length
        lens = np.array([min(c.size, n_samples) for c in clips], dtype=np.int32)
        width = max(int(lens.max()), 1)
        host = torch.zeros((len(clips), width), dtype=torch.float32).pin_memory() if torch.cuda.is_available() else None
        if host is None:
            raise RuntimeError("WhisperFeatureExtractor needs an MI355X GPU (no CPU fallback)")
        for i, c in enumerate(clips):
            host[i, : lens[i]] = torch.from_numpy(c[: lens[i]])
        dev = torch.device(device or self.device)
        pcm = host.to(dev, non_blocking=True)
        feats = logmel_whisper_device(pcm, torch.from_numpy(lens), int(lens.max()), n_frames, n_mels=self.feature_size)
        out = BatchFeature()
        if keep_on_device and return_tensors != "pt":
            raise ValueError("keep_on_device needs return_tensors='pt'")
        feats_cpu = feats if keep_on_device else feats.cpu()  # the reference's extractor returns host arrays
        want_mask = self.return_attention_mask if return_attention_mask is None else return_attention_mask
        if want_mask:
            mask = (np.arange(n_samples)[None, :] < lens[:, None]).astype(np.int32)[:, :: self.hop_length]
        if return_tensors == "pt":
            out["input_features"] = feats_cpu
            if want_mask:
                out["attention_mask"] = torch.from_numpy(mask)
        elif return_tensors == "np":
            out["input_features"] = feats_cpu.numpy()
            if want_mask:
                out["attention_mask"] = mask
        elif return_tensors is None:
            out["input_features"] = [f for f in feats_cpu.numpy()]
            if want_mask:
                out["attention_mask"] = [m for m in mask]
        else:
            raise ValueError("return_tensors must be None, 'np' or 'pt'")
        return out

    def pad(self, processed_features, return_tensors: Optional[str] = None, **kwargs) -> BatchFeature:
        """Stack already-extracted features (every row is [80, T] already, so padding is a pure stack)."""
        if isinstance(processed_features, dict):
            rows = processed_features["input_features"]
        else:
            rows = [f["input_features"] for f in processed_features]
        batch = stack_input_features(rows)
        if return_tensors == "np":
            return BatchFeature({"input_features": batch.numpy()})
        return BatchFeature({"input_features": batch})

    # -- helpers -----------------------------------------------------------------------------------------------
    @staticmethod
    def _to_clip_list(raw_speech) -> List[np.ndarray]:
        if isinstance(raw_speech, torch.Tensor):
            raw_speech = raw_speech.detach().cpu().numpy()
        if isinstance(raw_speech, np.ndarray):
            if raw_speech.ndim > 2:
                raise ValueError("Only mono-channel audio is supported for input to WhisperFeatureExtractor")
            clips = [raw_speech] if raw_speech.ndim == 1 else list(raw_speech)
        elif isinstance(raw_speech, (list, tuple)) and len(raw_speech) and isinstance(raw_speech[0], (np.ndarray, list, tuple, torch.Tensor)):
            clips = [c.detach().cpu().numpy() if isinstance(c, torch.Tensor) else c for c in raw_speech]
        else:
            clips = [raw_speech]
        out = []
        for c in clips:
            a = np.asarray(c, dtype=np.float32)
            if a.ndim != 1:
                raise ValueError("Only mono-channel audio is supported for input to WhisperFeatureExtractor")
            out.append(a)
        return out


class WhisperProcessor:
    """`feature_extractor` + optional `tokenizer`, callable like transformers.WhisperProcessor."""

    def __init__(self, feature_extractor: Optional[WhisperFeatureExtractor] = None, tokenizer: Any = None):
        self.feature_extractor = feature_extractor or WhisperFeatureExtractor()
        self.tokenizer = tokenizer

    def __call__(self, audio=None, sampling_rate: Optional[int] = None, text: Optional[Union[str, Sequence[str]]] = None, **kwargs):
        if audio is None and text is None:
            raise ValueError("You need to specify either an `audio` or `text` input to process.")
        out = BatchFeature()
        if audio is not None:
            out.update(self.feature_extractor(audio, sampling_rate=sampling_rate, **kwargs))
        if text is not None:
            if self.tokenizer is None:
                raise RuntimeError("WhisperProcessor was built without a tokenizer; pass tokenizer= to encode `text`")
            out["labels"] = self.tokenizer(text)["input_ids"]
        return out

    def batch_decode(self, *args, **kwargs):
        if self.tokenizer is None:
            raise RuntimeError("WhisperProcessor was built without a tokenizer")
        return self.tokenizer.batch_decode(*args, **kwargs)

    def decode(self, *args, **kwargs):
        if self.tokenizer is None:
            raise RuntimeError("WhisperProcessor was built without a tokenizer")
        return self.tokenizer.decode(*args, **kwargs)
