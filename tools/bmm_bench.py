#!/usr/bin/env python3
"""Batched-GEMM shapes of the absorbed cross-attention (native_decoder._AbsorbedCross) at B = 64, Whisper-small, L = 12, per forced tile.
    python tools/bmm_bench.py          (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mlx8_ws_audio_transformer_amd import _lib, native_decoder as nd

B, S, d, H, L = 64, 1500, 768, 12, 12
R, M = H * L, B * L
shapes = [("q~ = q_h Wk_h      ", H, M, d, 64), ("scores = q~ enc^T  ", B, R, S, d), ("ctx = P enc        ", B, R, d, S), ("a2 = ctx_h Wv_h^T  ", H, M, 64, d),
          ("d enc (2R deep)    ", B, S, d, 2 * R)]
torch.manual_seed(0)
for name, batch, m, n, k in shapes:
    a = torch.randn(batch, m, k, device="cuda")
    b = torch.randn(batch, n, k, device="cuda")
    pb = nd.PackedBatch(b)
    out = torch.empty(batch, m, n, device="cuda")
    line = "%s batch %3d M %5d N %5d K %5d:" % (name, batch, m, n, k)
    for tile in (0, 64, 128, 256):
        _lib.tuning_set("gemm_tile", tile)
        for _ in range(3):
            nd.bmm((a, 0, k, m * k), pb, m, (out, 0, n, m * n))
        _lib.prof_enable(True, ["gemm"])
        _lib.prof_collect("gemm")
        for _ in range(10):
            nd.bmm((a, 0, k, m * k), pb, m, (out, 0, n, m * n))
        torch.cuda.synchronize()
        ms = _lib.prof_collect("gemm")[0] / 10
        _lib.prof_enable(False)
        line += "  tile %3d: %6.1f us (%5.0f TF alg)" % (tile, ms * 1e3, 2.0 * batch * m * n * k / ms / 1e9)
    _lib.tuning_set("gemm_tile", 0)
    print(line, flush=True)
