import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mlx8_ws_audio_transformer_amd import _lib, ops
M = 48000
def t(n, k, tile):
    x = torch.randn(M, k, device="cuda"); w = torch.randn(n, k, device="cuda") * k ** -0.5
    _lib.tuning_set("gemm_tile", tile)
    ops.linear(x, w, None, "f16f8")
    _lib.prof_enable(True, ["gemm"]); _lib.prof_collect("gemm")
    for _ in range(8): ops.linear(x, w, None, "f16f8")
    ms, cnt, fl = _lib.prof_collect("gemm"); _lib.prof_enable(False)
    _lib.tuning_set("gemm_tile", 0)
    return ms / cnt
for name, n, npad, k in [("qkv", 1152, 1280, 384), ("out", 384, 512, 384), ("fc2", 384, 512, 1536), ("fc1", 1536, 1536, 384)]:
    a = t(n, k, 0); b = t(npad, k, 256); c = t(n, k, 128)
    print(f"{name}: N={n} auto {a*1e3:.1f} us | tile128 {c*1e3:.1f} us | padded N={npad} on 128x256 16x16 tiles {b*1e3:.1f} us")
