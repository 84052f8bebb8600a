#!/usr/bin/env python3
"""GEMM-only timing of the encoder's shapes through the C-ABI (HIP events inside libawt). GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx8_ws_audio_transformer_amd import _lib, ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 96000
shapes = [("qkv", M, 2304, 768), ("out", M, 768, 768), ("fc1", M, 3072, 768), ("fc2", M, 768, 3072), ("conv1", 2 * M, 768, 256)]
for prec in ["bf16x3", "fp16x3", "f16f8", "bf16"]:
    for name, m, n, k in shapes:
        x = torch.randn(m, k, device="cuda")
        w = torch.randn(n, k, device="cuda") * k ** -0.5
        y = ops.linear(x, w, None, prec)
        err = float((y[:256].double() - x[:256].double() @ w.double().t()).abs().max())
        _lib.prof_enable(True, ["gemm"]); _lib.prof_collect("gemm")
        for _ in range(6):
            ops.linear(x, w, None, prec)
        ms, cnt, fl = _lib.prof_collect("gemm"); _lib.prof_enable(False)
        print(f"{prec:7s} {name:6s} M={m} N={n} K={k}: {ms/cnt:8.3f} ms  {fl/ms/1e9:8.1f} TFLOP/s algorithmic   max|y - fp64| {err:.2e}")
        del x, w
