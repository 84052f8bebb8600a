"""Single-operator entry points of libawt (what the encoder is built from), as torch functions on device tensors.
Used by the per-kernel parity tests; each raises if the library or the GPU is missing."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

_TERMS = {"bf16": 1, "fp16": 2, "bf16x3": 3, "fp16x3": 4, "f16f8": 5, "f16f6": 6}   # common.h PREC_* (f16f6: experimental, `linear` only)


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, precision: str = "bf16x3") -> torch.Tensor:
    """y = x w^T + bias on the MFMA GEMM kernel; x [M, K], w [N, K] fp32 device tensors, N % 128 == 0, K % 64 == 0."""
    x, w = x.float().contiguous(), w.float().contiguous()
    M, K = x.shape
    N = w.shape[0]
    L = _lib.lib()
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    ws = _lib.workspace(L.awt_op_linear_workspace_bytes(M, N, K), x.device)
    b = bias.float().contiguous() if bias is not None else None
    with torch.cuda.device(x.device):
        _lib.check(L.awt_op_linear(_lib.ctx(x.device), _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), M, N, K,
                                   _TERMS[precision], _lib.ptr(ws), ws.numel(), _lib.stream_handle()))
    return y


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    x = x.float().contiguous()
    M, d = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().awt_op_layernorm(_lib.ctx(x.device), _lib.ptr(x), _lib.ptr(gamma.float().contiguous()),
                                               _lib.ptr(beta.float().contiguous()), _lib.ptr(y), M, d, eps, _lib.stream_handle()))
    return y


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, precision: str = "bf16x3") -> torch.Tensor:
    """softmax(q k^T) v for q (pre-scaled), k, v: [B, H, S, 64] fp32 -> [B, S, H * 64] fp32."""
    q, k, v = (t.float().contiguous() for t in (q, k, v))
    B, H, S, hd = q.shape
    if hd != 64:
        raise ValueError("head_dim must be 64")
    L = _lib.lib()
    o = torch.empty((B, S, H * 64), dtype=torch.float32, device=q.device)
    ws = _lib.workspace(L.awt_op_attention_workspace_bytes(B, H, S), q.device)
    with torch.cuda.device(q.device):
        _lib.check(L.awt_op_attention(_lib.ctx(q.device), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), B, H, S,
                                      _TERMS[precision], _lib.ptr(ws), ws.numel(), _lib.stream_handle()))
    return o
