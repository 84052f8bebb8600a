"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the Whisper audio encoder forward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The reference reaches this arithmetic through `WhisperForConditionalGeneration` / `WhisperModel.get_encoder()`
(/root/reference/AB/fineTune.py:131,199; /root/reference/.charles/music2midi/model.py:31-33,109-110); the
arithmetic itself is transformers' (pinned ==4.35.2 / 4.53.1, installed 5.15.0):

  WhisperEncoder.forward       HF:models/whisper/modeling_whisper.py:592-646
  WhisperEncoderLayer.forward  HF:...:379-413
  WhisperAttention.forward     HF:...:284-356  (q scaled by head_dim**-0.5 BEFORE q k^T, k_proj has no bias)
  eager_attention_forward      HF:...:215-238  (softmax without mask)
  sinusoids                    HF:...:55-64

It is restated here with plain torch CPU ops in fp32 (or fp64 for an error yardstick) and pinned
against the installed package by tools/make_golden.py -> tests/golden/encoder_*.npz.
The LoRA term has no reference implementation: "parity unpinned", defined here
as y = x W^T + b + (alpha / r) (x A^T) B^T.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _t(w, dtype):
    return w.to(dtype) if isinstance(w, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(w)).to(dtype)


def lora_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], a: Optional[torch.Tensor] = None,
                bm: Optional[torch.Tensor] = None, scale: float = 1.0) -> torch.Tensor:
    y = F.linear(x, w, b)
    if a is not None:
        y = y + scale * F.linear(F.linear(x, a), bm)
    return y


def encoder_forward(weights: Dict[str, np.ndarray], mel, heads: int, lora_scale: float = 0.0,
                    dtype=torch.float32, return_boundaries: bool = False):
    """mel [B, n_mels, T] -> last_hidden_state [B, T//2, d] (and optionally every layer boundary).

    `weights` uses HF state-dict keys (see mlx8-ws-audio-transformer_amd/weights.py); optional
    `<module>.lora_A` / `<module>.lora_B` entries add the LoRA term with `lora_scale` = alpha / r.
    """
    W = {k: _t(v, dtype) for k, v in weights.items()}
    x = _t(mel, dtype)
    d = W["conv1.weight"].shape[0]
    hd = d // heads
    S = W["embed_positions.weight"].shape[0]
    if x.shape[-1] != 2 * S:  # HF:modeling_whisper.py:612-616
        raise ValueError(f"Whisper expects the mel input features to be of length {2 * S}, but found {x.shape[-1]}.")
    bounds: List[torch.Tensor] = []

    h = F.gelu(F.conv1d(x, W["conv1.weight"], W["conv1.bias"], padding=1))
    h = F.gelu(F.conv1d(h, W["conv2.weight"], W["conv2.bias"], stride=2, padding=1))
    h = h.permute(0, 2, 1) + W["embed_positions.weight"]
    if return_boundaries:
        bounds.append(h.clone())

    def lin(name: str, inp: torch.Tensor, bias: bool = True) -> torch.Tensor:
        a = W.get(name + ".lora_A")
        return lora_linear(inp, W[name + ".weight"], W[name + ".bias"] if bias else None, a, W.get(name + ".lora_B"), lora_scale)

    n_layers = 1 + max(int(k.split(".")[1]) for k in W if k.startswith("layers."))
    B, S_, _ = h.shape
    for i in range(n_layers):
        p = f"layers.{i}."
        r = h
        y = F.layer_norm(h, (d,), W[p + "self_attn_layer_norm.weight"], W[p + "self_attn_layer_norm.bias"], 1e-5)
        q = (lin(p + "self_attn.q_proj", y) * hd ** -0.5).view(B, S_, heads, hd).transpose(1, 2)
        k = lin(p + "self_attn.k_proj", y, bias=False).view(B, S_, heads, hd).transpose(1, 2)
        v = lin(p + "self_attn.v_proj", y).view(B, S_, heads, hd).transpose(1, 2)
        att = torch.softmax(q @ k.transpose(2, 3), dim=-1) @ v
        att = att.transpose(1, 2).reshape(B, S_, d)
        h = r + lin(p + "self_attn.out_proj", att)
        r = h
        y = F.layer_norm(h, (d,), W[p + "final_layer_norm.weight"], W[p + "final_layer_norm.bias"], 1e-5)
        y = F.gelu(lin(p + "fc1", y))
        h = r + lin(p + "fc2", y)
        if return_boundaries:
            bounds.append(h.clone())
    out = F.layer_norm(h, (d,), W["layer_norm.weight"], W["layer_norm.bias"], 1e-5)
    if return_boundaries:
        return out, bounds
    return out


def error_norms(got, ref) -> Dict[str, float]:
    """The three norms SURVEY.md §0.5 asks every encoder parity report to state."""
    g = np.asarray(got, dtype=np.float64)
    r = np.asarray(ref, dtype=np.float64)
    diff = g - r
    return {
        "max_abs": float(np.abs(diff).max()),
        "mean_abs": float(np.abs(diff).mean()),
        "rel_l2": float(np.linalg.norm(diff) / max(np.linalg.norm(r), 1e-30)),
    }


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def encoder_forward_emulated(weights: Dict[str, np.ndarray], mel, heads: int, terms: int = 1) -> torch.Tensor:
    """Numerics model of the HIP design: bf16 MFMA operands (`terms`=1) or split-bf16 hi+lo operands with the
    three significant cross products (`terms`=3), fp32 accumulation, fp32 residual stream / LayerNorm / softmax.
    Used only to choose and document tolerances (DESIGN.md "Numerics"); not a parity oracle."""
    W = {k: _t(v, torch.float32) for k, v in weights.items()}

    def mm(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:  # a @ b with operand rounding
        ah, bh = bf16_round(a), bf16_round(b)
        if terms == 1:
            return ah @ bh
        al, bl = bf16_round(a - ah), bf16_round(b - bh)
        return ah @ bh + (ah @ bl + al @ bh)

    def lin(x, w, b=None):
        y = mm(x, w.t())
        return y if b is None else y + b

    x = _t(mel, torch.float32)
    d = W["conv1.weight"].shape[0]
    hd = d // heads
    B, C, T = x.shape

    def conv(xin, w, b, stride):  # im2col GEMM, K order (dt, c)
        Bn, Cn, Tn = xin.shape
        xp = F.pad(xin, (1, 1))
        cols = torch.stack([xp[:, :, dt: dt + Tn: 1] for dt in range(3)], dim=1)  # [B,3,C,T]
        cols = cols[..., ::stride].permute(0, 3, 1, 2).reshape(Bn, -1, 3 * Cn)
        wk = w.permute(0, 2, 1).reshape(w.shape[0], 3 * Cn)
        return lin(cols, wk, b)  # [B, T/stride, d]

    h = F.gelu(conv(x, W["conv1.weight"], W["conv1.bias"], 1)).permute(0, 2, 1)
    h = F.gelu(conv(h, W["conv2.weight"], W["conv2.bias"], 2)) + W["embed_positions.weight"]
    n_layers = 1 + max(int(k.split(".")[1]) for k in W if k.startswith("layers."))
    S_ = h.shape[1]
    for i in range(n_layers):
        p = f"layers.{i}."
        y = F.layer_norm(h, (d,), W[p + "self_attn_layer_norm.weight"], W[p + "self_attn_layer_norm.bias"], 1e-5)
        q = ((lin(y, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"])) * hd ** -0.5)
        k = lin(y, W[p + "self_attn.k_proj.weight"])
        v = lin(y, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"])
        q, k, v = (t.view(B, S_, heads, hd).transpose(1, 2) for t in (q, k, v))
        s = mm(q, k.transpose(2, 3))
        pm = torch.softmax(s, dim=-1)
        att = mm(pm, v).transpose(1, 2).reshape(B, S_, d)
        h = h + lin(att, W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"])
        y = F.layer_norm(h, (d,), W[p + "final_layer_norm.weight"], W[p + "final_layer_norm.bias"], 1e-5)
        y = F.gelu(lin(y, W[p + "fc1.weight"], W[p + "fc1.bias"]))
        h = h + lin(y, W[p + "fc2.weight"], W[p + "fc2.bias"])
    return F.layer_norm(h, (d,), W["layer_norm.weight"], W["layer_norm.bias"], 1e-5)
