#!/usr/bin/env python3
"""Attention-only timing at the bench shape through the C-ABI (HIP events inside libawt). GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx8_ws_audio_transformer_amd import _lib, ops

B, H, S = 64, 12, 1500
for prec in ["bf16x3", "bf16"]:
    q, k, v = (torch.randn(B, H, S, 64, device="cuda") for _ in range(3))
    ops.attention(q, k, v, prec)
    _lib.prof_enable(True, ["attention"]); _lib.prof_collect("attention")
    for _ in range(5):
        ops.attention(q, k, v, prec)
    ms, cnt, fl = _lib.prof_collect("attention"); _lib.prof_enable(False)
    print(f"{prec:7s} attention B={B} H={H} S={S}: {ms/cnt:8.3f} ms  {fl/ms/1e9:8.1f} TFLOP/s algorithmic")
