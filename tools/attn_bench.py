#!/usr/bin/env python3
"""Attention-only timing at the bench shape through the C-ABI (HIP events inside libawt). GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx8_ws_audio_transformer_amd import _lib, ops

B, H, S = 64, 12, 1500
q, k, v = (torch.randn(B, H, S, 64, device="cuda") for _ in range(3))
q *= 0.35
ref = (torch.softmax(q[:2].double() @ k[:2].double().transpose(2, 3), dim=-1) @ v[:2].double()).transpose(1, 2).reshape(2, S, H * 64)
for prec, shape in [("bf16x3", 0), ("fp16x3", 0), ("f16f8", 1), ("f16f8", 2), ("f16f8", 3), ("f16f8", 4), ("f16f8", 5), ("f16f8", 6), ("f16f8", 7), ("f16f8", 0), ("bf16", 0)]:
    _lib.tuning_set("attn_shape", shape)
    o = ops.attention(q, k, v, prec)
    _lib.prof_enable(True, ["attention"]); _lib.prof_collect("attention")
    for _ in range(8):
        ops.attention(q, k, v, prec)
    ms, cnt, fl = _lib.prof_collect("attention"); _lib.prof_enable(False)
    print(f"{prec:7s} shape {shape} attention B={B} H={H} S={S}: {ms/cnt:8.3f} ms  {fl/ms/1e9:8.1f} TFLOP/s algorithmic   max|o - fp64| {float((o[:2].double() - ref).abs().max()):.2e}")
_lib.tuning_set("attn_shape", 0)
