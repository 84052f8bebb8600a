#!/usr/bin/env python3
"""f16f8 GEMM at the encoder's shapes per block-tile configuration (awt_tuning_set "gemm_tile": 128 = 128x128, 256 = 128x256 on 4 waves,
512 = 256x256 on 8 waves), with the result checked against the default configuration. GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx8_ws_audio_transformer_amd import _lib, ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 96000
shapes = [("qkv", M, 2304, 768), ("out", M, 768, 768), ("fc1", M, 3072, 768), ("fc2", M, 768, 3072)]
for name, m, n, k in shapes:
    x = torch.randn(m, k, device="cuda")
    w = torch.randn(n, k, device="cuda") * k ** -0.5
    b = torch.randn(n, device="cuda")
    ref = None
    for tile in (256, 128, 512, 256, 512):
        _lib.tuning_set("gemm_tile", tile)
        y = ops.linear(x, w, b, "f16f8")
        if ref is None:
            ref = y
        _lib.prof_enable(True, ["gemm"]); _lib.prof_collect("gemm")
        for _ in range(10):
            ops.linear(x, w, b, "f16f8")
        ms, cnt, fl = _lib.prof_collect("gemm"); _lib.prof_enable(False)
        print(f"{name:4s} tile {tile}: {ms/cnt:7.3f} ms  {fl/ms/1e9:7.1f} TFLOP/s   max|y - y(256)| {float((y - ref).abs().max()):.1e}", flush=True)
    del x, w
_lib.tuning_set("gemm_tile", 0)
