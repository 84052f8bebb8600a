#!/usr/bin/env python3
"""Tile-order knob sweep (row panels per group) on the encoder's GEMM shapes. GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx8_ws_audio_transformer_amd import _lib, ops
M = 96000
shapes = [("qkv", M, 2304, 768), ("out", M, 768, 768), ("fc1", M, 3072, 768), ("fc2", M, 768, 3072)]
prec = sys.argv[1] if len(sys.argv) > 1 else "f16f8"
data = {n: (torch.randn(m, k, device="cuda"), torch.randn(nn, k, device="cuda") * k ** -0.5) for n, m, nn, k in shapes}
for rep in range(2):
    for gm in (4, 6, 8, 12, 16, 24):
        _lib.tuning_set("gemm_gm", gm)
        tot = 0.0; line = []
        for name, m, n, k in shapes:
            x, w = data[name]
            ops.linear(x, w, None, prec)
            _lib.prof_enable(True, ["gemm"]); _lib.prof_collect("gemm")
            for _ in range(5):
                ops.linear(x, w, None, prec)
            ms, cnt, fl = _lib.prof_collect("gemm"); _lib.prof_enable(False)
            tot += ms / cnt; line.append(f"{name} {ms/cnt:.3f}")
        print(f"rep {rep} {prec} GM={gm:2d}: " + "  ".join(line) + f"  | layer {tot:.3f} ms", flush=True)
_lib.tuning_set("gemm_gm", 0)
