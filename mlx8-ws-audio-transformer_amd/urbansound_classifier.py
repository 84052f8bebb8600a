"""`TransformerUrbanSound8KClassifier` (/root/reference/.charles/spectrogram.py:944-1057) with its inference path on libawt.

Same constructor, same parameter names as the reference module (`input_proj`, `cls_token`, `pos_embed`,
`encoder.layers.{i}.self_attn.in_proj_weight / in_proj_bias / out_proj`, `linear1`, `linear2`, `norm1`, `norm2`, `norm`,
`head.0`, `head.3`), so a checkpoint written by the reference's `train_transformer` loads with `load_state_dict`.

`forward` / `get_feature_embeddings` in eval mode run the linears, the multi-head attention and the LayerNorms through the
C-ABI's single-operator entry points (`awt_op_linear`, `awt_op_attention`, `awt_op_layernorm`): a second consumer of the
encoder's kernels at a different shape -- d = 128, 4 heads of 32 (zero-padded to the kernel's head_dim 64, which leaves
q k^T and the first 32 dims of P v unchanged), S = 127 / 502 + CLS, **post**-LN.  The element-wise GELU / ReLU between two
native calls and the CLS / positional adds are torch tensor ops on the device.

Training (`model.train()`, `train_transformer` spectrogram.py:1059-1130) runs on the same library: each operator is a
`torch.autograd.Function` whose forward AND backward are libawt calls --
  * linears: y = x W^T on the MFMA GEMM; dx = dy W and dW = dy^T x are the same GEMM with the operands transposed (the
    contraction of dW runs over the B (T + 1) rows), db = `awt_op_column_sums`;
  * attention: the fp32 row kernels `awt_op_attention_small` / `_backward` (heads zero-padded to 64, q pre-multiplied by
    sqrt(64 / head_dim) because the kernel's score scale is 64^-1/2);
  * LayerNorm: `awt_op_layernorm`, `awt_op_layernorm_backward` (dx) and `awt_op_layernorm_param_grad` (dgamma, dbeta);
  * GELU: `awt_op_gelu` / `awt_op_gelu_backward`; loss: `awt_op_cross_entropy` (`native_cross_entropy`).
Dropout masks, ReLU in the head, the residual adds, the CLS concat and the head padding are element-wise torch tensor ops
under autograd; the optimizer is `torch.optim.Adam` like the reference's.  There is no torch fallback for any operator above:
without libawt the first call raises.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, native_decoder as nd, ops
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


# ------------------------------------------------------------------------------------------------ trainable operators
def _column_sums(a: torch.Tensor) -> torch.Tensor:
    M, d = a.shape
    out = torch.empty(d, dtype=torch.float32, device=a.device)
    L = _lib.lib()
    ws = _lib.workspace(L.awt_op_param_grad_workspace_bytes(M, d), a.device)
    with torch.cuda.device(a.device):
        _lib.check(L.awt_op_column_sums(_lib.ctx(a.device), _lib.ptr(a), _lib.ptr(out), M, d, _lib.ptr(ws), ws.numel(), _lib.stream_handle()))
    return out


class _Linear(torch.autograd.Function):
    """y [M, N] = x [M, K] w[N, K]^T + b with all three gradients on libawt."""

    @staticmethod
    def forward(ctx, x, w, b, precision):
        ctx.save_for_backward(x, w)
        ctx.precision, ctx.has_bias = precision, b is not None
        return native_linear(x, w, b, precision)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = native_linear(dy, w.t().contiguous(), None, ctx.precision) if ctx.needs_input_grad[0] else None
        dw = native_linear(dy.t().contiguous(), x.t().contiguous(), None, ctx.precision)          # [N, M] x [K, M]^T
        db = _column_sums(dy) if ctx.has_bias else None
        return dx, dw, db, None


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        ctx.save_for_backward(x, g)
        ctx.eps = eps
        return nd.layernorm(x, g, b, eps)

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dy = dy.contiguous()
        M, d = x.shape
        dx = nd.layernorm_backward(dy, x, g, None, ctx.eps)
        dg, db = torch.empty_like(g), torch.empty_like(g)
        L = _lib.lib()
        ws = _lib.workspace(L.awt_op_param_grad_workspace_bytes(M, d), x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.awt_op_layernorm_param_grad(_lib.ctx(x.device), _lib.ptr(dy), _lib.ptr(x), _lib.ptr(dg), _lib.ptr(db), M, d, ctx.eps,
                                                     _lib.ptr(ws), ws.numel(), _lib.stream_handle()))
        return dx, dg, db, None


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return nd.gelu(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return nd.gelu_backward(x, dy.contiguous())


class _Attention(torch.autograd.Function):
    """qkv [B * S, 3 * H * 64] (q | k | v blocks of H heads of 64) -> o [B * S, H * 64]; score scale 64^-1/2 inside the kernel."""

    @staticmethod
    def forward(ctx, qkv, B, H, S):
        ld, w = qkv.shape[1], H * 64
        o, lse = nd.attention_small((qkv, 0), ld, (qkv, w), ld, (qkv, 2 * w), ld, B, H, S, S, False, 0)
        ctx.save_for_backward(qkv, o, lse)
        ctx.shape = (B, H, S)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        B, H, S = ctx.shape
        ld, w = qkv.shape[1], H * 64
        dqkv = torch.empty_like(qkv)
        nd.attention_small_backward((qkv, 0), ld, (qkv, w), ld, (qkv, 2 * w), ld, o, do.contiguous(), lse, (dqkv, 0), (dqkv, w), (dqkv, 2 * w),
                                    B, H, S, S, False, 0)
        return dqkv, None, None, None


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        loss, dlogits = nd.cross_entropy(logits.contiguous(), labels, logits.shape[1])
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None


def native_cross_entropy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """`torch.nn.CrossEntropyLoss()(logits, labels)` (spectrogram.py:1107) on `awt_op_cross_entropy`, differentiable."""
    return _CrossEntropy.apply(logits.float(), labels)


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
            return self._features_train(x)
        x = x.to(self.input_proj.weight.device, torch.float32).transpose(1, 2)               # [B, T, n_mels]
        B, T, _ = x.shape
        x = native_linear(x, self.input_proj.weight, self.input_proj.bias, self.precision)
        x = torch.cat([self.cls_token.expand(B, -1, -1), x], dim=1)
        x = x + self._positions(T, x.device)
        for layer in self.encoder.layers:                                                    # post-LN (norm_first=False)
            x = self._layernorm(layer.norm1, x + self._attention(layer, x))
            ff = native_linear(F.gelu(native_linear(x, layer.linear1.weight, layer.linear1.bias, self.precision)),
                               layer.linear2.weight, layer.linear2.bias, self.precision)
            x = self._layernorm(layer.norm2, x + ff)
        return self._layernorm(self.norm, x)[:, 0]                                           # CLS token

    def _positions(self, T: int, device) -> torch.Tensor:
        if self.pos_embed is None or self.n_frames != T:                                     # spectrogram.py:1017-1021
            self.n_frames = T
            self.pos_embed = nn.Parameter(torch.zeros(1, T + 1, self.dim, device=device))
            nn.init.trunc_normal_(self.pos_embed, std=0.02)
        return self.pos_embed

    # ---------------------------------------------------------------------------------------------------- training path
    def _features_train(self, x: torch.Tensor) -> torch.Tensor:
        """The same network with every operator a libawt autograd node (module docstring).  Dropout follows the reference's placement
        (after the positional add, after attention, inside and after the MLP, in the head); the attention-probability dropout
        that `nn.MultiheadAttention` adds is not applied -- the native attention kernel has no dropout."""
        P, drop = self.precision, self.dropout
        if P not in ("bf16", "bf16x3"):
            raise ValueError("training runs the GEMMs in bf16 / bf16x3 operand planes (gradients need fp32's exponent range)")
        x = x.to(self.input_proj.weight.device, torch.float32).transpose(1, 2)               # [B, T, n_mels]
        B, T, _ = x.shape
        S, d, H = T + 1, self.dim, self.heads
        hd = d // H
        x = _Linear.apply(x.reshape(B * T, -1).contiguous(), self.input_proj.weight, self.input_proj.bias, P).reshape(B, T, d)
        x = torch.cat([self.cls_token.expand(B, -1, -1), x], dim=1)
        x = drop(x + self._positions(T, x.device)).reshape(B * S, d)
        for layer in self.encoder.layers:
            a = layer.self_attn
            qkv = _Linear.apply(x, a.in_proj_weight, a.in_proj_bias, P).reshape(B * S, 3, H, hd)
            scale = qkv.new_tensor([math.sqrt(64.0 / hd), 1.0, 1.0]).reshape(1, 3, 1, 1)
            qkv = F.pad(qkv * scale, (0, 64 - hd)).reshape(B * S, 3 * H * 64)
            o = _Attention.apply(qkv, B, H, S).reshape(B * S, H, 64)[..., :hd].reshape(B * S, d)
            o = _Linear.apply(o.contiguous(), a.out_proj.weight, a.out_proj.bias, P)
            x = _LayerNorm.apply((x + layer.dropout1(o)).contiguous(), layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
            h = layer.dropout(_Gelu.apply(_Linear.apply(x, layer.linear1.weight, layer.linear1.bias, P)))
            ff = _Linear.apply(h.contiguous(), layer.linear2.weight, layer.linear2.bias, P)
            x = _LayerNorm.apply((x + layer.dropout2(ff)).contiguous(), layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
        x = _LayerNorm.apply(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x.reshape(B, S, d)[:, 0]

    # ---------------------------------------------------------------------------------------------------- reference surface
    def get_feature_embeddings(self, x: torch.Tensor) -> torch.Tensor:
        """[B, n_mels, n_frames] -> CLS features [B, dim] (spectrogram.py:1040-1057)."""
        if self.training:
            return self._features(x)
        with torch.no_grad():
            return self._features(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[B, n_mels, n_frames] -> logits [B, n_classes] (spectrogram.py:997-1038).  eval(): inference kernels under no_grad;
        train(): differentiable, every operator's backward in libawt."""
        if self.training:
            cls = self._features(x).contiguous()
            P = self.precision
            h = self.head[2](F.relu(_Linear.apply(cls, self.head[0].weight, self.head[0].bias, P)))
            return _Linear.apply(h.contiguous(), self.head[3].weight, self.head[3].bias, P)
        with torch.no_grad():
            cls = self._features(x)
            h = F.relu(native_linear(cls, self.head[0].weight, self.head[0].bias, self.precision))
            return native_linear(h, self.head[3].weight, self.head[3].bias, self.precision)


def train_transformer(train_loader, model: Optional[TransformerUrbanSound8KClassifier] = None, epochs: int = 1, lr: float = 1e-3,
                      weight_decay: float = 0.0, n_classes: int = 10, n_mels: int = N_MELS, device="cuda", log=None):
    """The reference's training loop (spectrogram.py:1059-1130) over the native operators: Adam, `native_cross_entropy`, one optimizer
    step per batch; returns (model, per-epoch mean loss).  Data loading, evaluation metrics, wandb and checkpoint naming stay with the
    caller (outside SURVEY.md section 8); `train_loader` yields (xb [B, n_mels, n_frames], yb [B])."""
    if model is None:
        model = TransformerUrbanSound8KClassifier(n_classes=n_classes, n_mels=n_mels).to(device)
    # Like the reference, Adam is built before the first forward, i.e. before `pos_embed` exists (spectrogram.py:1017-1021, 1104): the
    # positional table of a freshly constructed model keeps its initial values; a model that already ran a forward trains it too.
    optimizer = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)
    losses = []
    for epoch in range(epochs):
        model.train()
        total, seen = 0.0, 0
        for xb, yb in train_loader:
            xb, yb = xb.to(device), yb.to(device)
            optimizer.zero_grad()
            logits = model(xb)
            loss = native_cross_entropy(logits, yb)
            loss.backward()
            optimizer.step()
            total += float(loss.detach()) * xb.size(0)
            seen += xb.size(0)
        losses.append(total / max(seen, 1))
        if log is not None:
            log(f"Epoch {epoch + 1}: Train loss={losses[-1]:.4f}")
    return model, losses
