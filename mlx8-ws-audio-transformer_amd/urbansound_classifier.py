"""`TransformerUrbanSound8KClassifier` (/root/reference/.charles/spectrogram.py:944-1057) with its inference path on libawt.

Same constructor, same parameter names as the reference module (`input_proj`, `cls_token`, `pos_embed`,
`encoder.layers.{i}.self_attn.in_proj_weight / in_proj_bias / out_proj`, `linear1`, `linear2`, `norm1`, `norm2`, `norm`,
`head.0`, `head.3`), so a checkpoint written by the reference's `train_transformer` loads with `load_state_dict`.

`forward` / `get_feature_embeddings` in eval mode run the linears, the multi-head attention and the LayerNorms through the
C-ABI's single-operator entry points (`awt_op_linear`, `awt_op_attention`, `awt_op_layernorm`): a second consumer of the
encoder's kernels at a different shape -- d = 128, 4 heads of 32 (zero-padded to the kernel's head_dim 64, which leaves
q k^T and the first 32 dims of P v unchanged), S = 127 / 502 + CLS, **post**-LN.  The element-wise GELU / ReLU between two
native calls and the CLS / positional adds are torch tensor ops on the device.  Training this model natively is outside the
scope: in training mode `forward` raises instead of falling back.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .urbansound import N_MELS

TRANSFORMER_DIM, TRANSFORMER_HEADS, TRANSFORMER_LAYERS, TRANSFORMER_DROPOUT, TRANSFORMER_MLP_DIM = 128, 4, 2, 0.1, 256   # spectrogram.py:70-74


def _pad_rows(w: torch.Tensor, b: Optional[torch.Tensor], multiple: int = 128):
    n = w.shape[0]
    n_pad = (n + multiple - 1) // multiple * multiple
    if n_pad == n:
        return w, b, n
    w = F.pad(w, (0, 0, 0, n_pad - n))
    b = F.pad(b, (0, n_pad - n)) if b is not None else None
    return w, b, n


def native_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], precision: str) -> torch.Tensor:
    """x [..., K] @ w[N, K]^T + b on the MFMA GEMM: K zero-padded to a multiple of 64, N to a multiple of 128."""
    lead, K = x.shape[:-1], x.shape[-1]
    k_pad = (K + 63) // 64 * 64
    x2 = x.reshape(-1, K)
    if k_pad != K:
        x2, w = F.pad(x2, (0, k_pad - K)), F.pad(w, (0, k_pad - K))
    w, b, n = _pad_rows(w, b)
    return ops.linear(x2, w, b, precision)[:, :n].reshape(*lead, n)


class TransformerUrbanSound8KClassifier(nn.Module):
    def __init__(self, n_classes: int = 10, n_mels: int = N_MELS, n_frames: Optional[int] = None, dim: int = TRANSFORMER_DIM,
                 depth: int = TRANSFORMER_LAYERS, heads: int = TRANSFORMER_HEADS, mlp_dim: int = TRANSFORMER_MLP_DIM,
                 dropout: float = TRANSFORMER_DROPOUT, precision: str = "bf16x3"):
        super().__init__()
        if dim % heads or dim // heads > 64:
            raise ValueError("head_dim must divide dim and be <= 64 (the native attention kernel's head_dim)")
        self.n_mels, self.dim, self.heads, self.precision = n_mels, dim, heads, precision
        self.input_proj = nn.Linear(n_mels, dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.pos_embed = None                      # created on first use, as the reference does (spectrogram.py:1017-1021)
        self.n_frames = n_frames
        layer = nn.TransformerEncoderLayer(d_model=dim, nhead=heads, dim_feedforward=mlp_dim, dropout=dropout, activation="gelu",
                                           batch_first=True)
        self.encoder = nn.TransformerEncoder(layer, num_layers=depth)
        self.dropout = nn.Dropout(dropout)
        self.norm = nn.LayerNorm(dim)
        self.head = nn.Sequential(nn.Linear(dim, mlp_dim), nn.ReLU(), nn.Dropout(dropout), nn.Linear(mlp_dim, n_classes))

    # ---------------------------------------------------------------------------------------------------- native pieces
    def _attention(self, layer, x: torch.Tensor) -> torch.Tensor:
        B, S, d = x.shape
        H, hd = self.heads, d // self.heads
        a = layer.self_attn
        qkv = native_linear(x, a.in_proj_weight, a.in_proj_bias, self.precision)           # [B, S, 3 d]
        q, k, v = (t.reshape(B, S, H, hd).transpose(1, 2) for t in qkv.split(d, dim=-1))     # [B, H, S, hd]
        q = q * (hd ** -0.5)                                                                 # ops.attention takes a pre-scaled q
        if hd < 64:
            q, k, v = (F.pad(t, (0, 64 - hd)) for t in (q, k, v))
        o = ops.attention(q, k, v, self.precision).reshape(B, S, H, 64)[..., :hd].reshape(B, S, d)
        return native_linear(o, a.out_proj.weight, a.out_proj.bias, self.precision)

    def _layernorm(self, ln: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
        return ops.layernorm(x.reshape(-1, x.shape[-1]), ln.weight, ln.bias, ln.eps).reshape(x.shape)

    def _features(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise NotImplementedError("the native TransformerUrbanSound8KClassifier implements inference only: call .eval() "
                                      "(training this model is outside the native scope, there is no torch fallback)")
        x = x.to(self.input_proj.weight.device, torch.float32).transpose(1, 2)               # [B, T, n_mels]
        B, T, _ = x.shape
        x = native_linear(x, self.input_proj.weight, self.input_proj.bias, self.precision)
        x = torch.cat([self.cls_token.expand(B, -1, -1), x], dim=1)
        if self.pos_embed is None or self.n_frames != T:
            self.n_frames = T
            self.pos_embed = nn.Parameter(torch.zeros(1, T + 1, self.dim, device=x.device))
            nn.init.trunc_normal_(self.pos_embed, std=0.02)
        x = x + self.pos_embed
        for layer in self.encoder.layers:                                                    # post-LN (norm_first=False)
            x = self._layernorm(layer.norm1, x + self._attention(layer, x))
            ff = native_linear(F.gelu(native_linear(x, layer.linear1.weight, layer.linear1.bias, self.precision)),
                               layer.linear2.weight, layer.linear2.bias, self.precision)
            x = self._layernorm(layer.norm2, x + ff)
        return self._layernorm(self.norm, x)[:, 0]                                           # CLS token

    # ---------------------------------------------------------------------------------------------------- reference surface
    @torch.no_grad()
    def get_feature_embeddings(self, x: torch.Tensor) -> torch.Tensor:
        """[B, n_mels, n_frames] -> CLS features [B, dim] (spectrogram.py:1040-1057)."""
        return self._features(x)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[B, n_mels, n_frames] -> logits [B, n_classes] (spectrogram.py:997-1038)."""
        cls = self._features(x)
        h = F.relu(native_linear(cls, self.head[0].weight, self.head[0].bias, self.precision))
        return native_linear(h, self.head[3].weight, self.head[3].bias, self.precision)
