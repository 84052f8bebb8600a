"""MI355X-native log-mel front-end + Whisper-style audio encoder (hot path of
AdamBeedell/MLX8-WS-Audio-Transformer), behind the reference's own Python call surface.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed).  Compute: hand-written
HIP for gfx950 in csrc/, reached only through the C-ABI in include/awt.h (libawt.so, loaded by _lib.py).
There is no CPU fallback: every operator raises if the library is missing.
"""
from .weights import (CONFIGS, EncoderConfig, LoraSpec, config, init_encoder_weights,  # noqa: F401
                      init_lora_weights, weights_digest)
