#!/usr/bin/env python3
"""Times the pipelined f16f8 attention kernel (shape 4) of whatever libawt AWT_LIB points at. GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx8_ws_audio_transformer_amd import _lib, ops
B, H, S = 64, 12, 1500
q, k, v = (torch.randn(B, H, S, 64, device="cuda") for _ in range(3)); q *= 0.35
for shape in (6, 4):
    _lib.tuning_set("attn_shape", shape)
    ops.attention(q, k, v, "f16f8")
    _lib.prof_enable(True, ["attention"]); _lib.prof_collect("attention")
    for _ in range(8):
        ops.attention(q, k, v, "f16f8")
    ms, cnt, fl = _lib.prof_collect("attention"); _lib.prof_enable(False)
    print(f"shape {shape}: {ms/cnt:8.3f} ms")
