"""`WhisperAudioEncoder`: waveform -> log-mel -> encoder hidden states, the reference's audio tower
(/root/reference/.charles/music2midi/model.py:23-123), batched and on-device.

The reference pads the waveforms to a common length, then loops PER SAMPLE through the HF processor (CPU) and the
encoder (B = 1 launches) and concatenates (model.py:96-121).  Here the whole batch goes to the GPU once as PCM and
`awt_audio_encode` runs log-mel + encoder for all clips; the result is the same [B, 1500, d] tensor.
"""
from __future__ import annotations

from typing import Optional, List, Union

import numpy as np
import torch
import torch.nn as nn

from .encoder import NativeWhisperEncoder
from .feature_extraction import WhisperProcessor
from .weights import EncoderConfig, config as named_config


class WhisperAudioEncoder(nn.Module):
    def __init__(self, model_name: Union[str, EncoderConfig] = "base", freeze_encoder: bool = True, precision: Optional[str] = None,
                 device: str = "cuda", state_dict=None):
        super().__init__()
        cfg = model_name if isinstance(model_name, EncoderConfig) else named_config(str(model_name).split("whisper-")[-1])
        self.processor = WhisperProcessor()
        self.encoder = NativeWhisperEncoder(cfg, precision=precision, device=device)
        if state_dict is not None:
            self.encoder.load_state_dict(state_dict)
        if freeze_encoder:
            for p in self.encoder.parameters():
                p.requires_grad = False
        self.encoder.eval()

    def forward(self, waveforms: Union[torch.Tensor, List[np.ndarray]], sampling_rate: int) -> torch.Tensor:
        if sampling_rate != 16000:  # what the processor call raises in the reference (feature_extraction_whisper.py:265-271)
            raise ValueError(f"WhisperFeatureExtractor was trained using a sampling rate of 16000, not {sampling_rate}.")
        if isinstance(waveforms, list):
            clips = []
            for wf in waveforms:
                t = torch.from_numpy(wf).float() if isinstance(wf, np.ndarray) else wf.float()
                if t.dim() > 1:
                    t = t.mean(dim=0)          # model.py:58-61
                clips.append(t)
            max_len = max(t.shape[0] for t in clips)
            batch = torch.zeros((len(clips), max_len), dtype=torch.float32)
            for i, t in enumerate(clips):      # zero-pad to the batch maximum (model.py:63-76)
                batch[i, : t.shape[0]] = t
            waveforms = batch
        if waveforms.dim() == 1:
            waveforms = waveforms.unsqueeze(0)
        elif waveforms.dim() > 2:
            waveforms = waveforms.mean(dim=-1)  # model.py:82-84
        n = min(waveforms.shape[1], 480000)
        pcm = waveforms[:, :n].to(self.encoder.device, torch.float32).contiguous()
        with torch.no_grad():
            hidden = self.encoder.encode_pcm(pcm)
        return hidden.to(dtype=next(self.encoder.parameters()).dtype)  # model.py:113-121
