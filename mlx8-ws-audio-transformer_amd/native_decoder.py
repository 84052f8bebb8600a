"""`NativeWhisperDecoder`: the decoder half of `WhisperForConditionalGeneration.forward` on libawt (scope row f1).

Reference: /root/reference/AB/fineTune.py:131 loads `WhisperForConditionalGeneration`, `trainer.train()` (:186-199) calls its
forward with `input_features` and `labels`: shift labels right, encoder, decoder (pre-LN layers of causal self-attention,
cross-attention over the 1500 encoder positions, GELU MLP), tied vocabulary projection, cross-entropy with ignore_index -100
(HF:modeling_whisper.py:416-507, 649-797, 994-1100).

Everything after the encoder runs in libawt here (include/awt.h "Decoder-side operators"): the linears on the MFMA GEMM against
weights packed once (`awt_weight`), attention in the fp32 row kernels, embedding / LayerNorm / GELU / cross-entropy as row
kernels.  The decoder weights are frozen (only the encoder's adapters train), so the backward pass carries ONE gradient, d(loss) /
d(encoder hidden states), and the whole decoder + loss is a single autograd node (`_DecoderLoss`): no autograd graph, no per-layer
tensors beyond the activations the native backward needs, and the twelve layers' cross-attention key / value gradients are written
side by side into one buffer that a single GEMM turns into d(encoder hidden states).

Parameter names are HF's (`WhisperDecoder.state_dict()` loads unchanged); `forward` has `finetune.WhisperDecoder`'s signature, so
`greedy_decode` / `generate` run on it as well (incremental decoding with a self-attention cache).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib

_PREC = {"bf16": 1, "bf16x3": 3}


class PackedLinear:
    """A frozen nn.Linear packed once into libawt's fragment-major operand planes (`awt_weight`)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], precision: str = "bf16x3", backward: bool = True):
        w = weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        self.N, self.K = w.shape
        self.device = w.device
        out = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().awt_weight_create(_lib.ctx(self.device), _lib.ptr(w), _lib.ptr(b), self.N, self.K, _PREC[precision], int(backward),
                                                    _lib.stream_handle(), C.byref(out)))
        self.handle = out.value
        self.Np = int(_lib.lib().awt_weight_padded_rows(self.handle))

    def forward(self, x: torch.Tensor, resid: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """y [M, Np] = x [M, K] W^T + b (+ resid)."""
        L = _lib.lib()
        M = x.shape[0]
        y = torch.empty((M, self.Np), dtype=torch.float32, device=x.device) if out is None else out
        ws = _lib.workspace(L.awt_linear_workspace_bytes(self.handle, M, 0), x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.awt_linear_forward(_lib.ctx(x.device), self.handle, _lib.ptr(x), _lib.ptr(resid), _lib.ptr(y), M, _lib.ptr(ws), ws.numel(),
                                            _lib.stream_handle()))
        return y

    def backward_input(self, dy: torch.Tensor) -> torch.Tensor:
        """dx [M, K] = dy [M, Np] W."""
        L = _lib.lib()
        M = dy.shape[0]
        dx = torch.empty((M, self.K), dtype=torch.float32, device=dy.device)
        ws = _lib.workspace(L.awt_linear_workspace_bytes(self.handle, M, 1), dy.device)
        with torch.cuda.device(dy.device):
            _lib.check(L.awt_linear_backward_input(_lib.ctx(dy.device), self.handle, _lib.ptr(dy), _lib.ptr(dx), M, _lib.ptr(ws), ws.numel(),
                                                   _lib.stream_handle()))
        return dx

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().awt_weight_destroy(self.handle)
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------ thin op wrappers
def _ctx(t):
    return _lib.ctx(t.device)


def layernorm(x, g, b, eps=1e-5):
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().awt_op_layernorm(_ctx(x), _lib.ptr(x), _lib.ptr(g), _lib.ptr(b), _lib.ptr(y), x.shape[0], x.shape[1], eps, _lib.stream_handle()))
    return y


def layernorm_backward(dy, x, g, dres=None, eps=1e-5):
    dx = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().awt_op_layernorm_backward(_ctx(x), _lib.ptr(dy), _lib.ptr(x), _lib.ptr(g), _lib.ptr(dres), _lib.ptr(dx), x.shape[0], x.shape[1],
                                                        eps, _lib.stream_handle()))
    return dx


def gelu(x):
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().awt_op_gelu(_ctx(x), _lib.ptr(x), _lib.ptr(y), x.numel(), _lib.stream_handle()))
    return y


def gelu_backward(x, dy):
    dx = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().awt_op_gelu_backward(_ctx(x), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dx), x.numel(), _lib.stream_handle()))
    return dx


def attention_small(q, ldq, k, ldk, v, ldv, B, H, Lq, Sk, causal, causal_off, want_lse=True):
    """q / k / v: (tensor, column offset in elements) pairs over row-major fp32 matrices; returns o [B * Lq, H * 64] and lse [B, H, Lq]."""
    (qt, qo), (kt, ko), (vt, vo) = q, k, v
    o = torch.empty((B * Lq, H * 64), dtype=torch.float32, device=qt.device)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=qt.device) if want_lse else None
    with torch.cuda.device(qt.device):
        _lib.check(_lib.lib().awt_op_attention_small(_ctx(qt), qt.data_ptr() + 4 * qo, ldq, kt.data_ptr() + 4 * ko, ldk, vt.data_ptr() + 4 * vo, ldv,
                                                     _lib.ptr(o), H * 64, _lib.ptr(lse), B, H, Lq, Sk, int(causal), int(causal_off), _lib.stream_handle()))
    return o, lse


def attention_small_backward(q, ldq, k, ldk, v, ldv, o, dout, lse, dq, dk, dv, B, H, Lq, Sk, causal, causal_off):
    """Writes dq / dk / dv at (tensor, column offset) destinations laid out like q / k / v."""
    (qt, qo), (kt, ko), (vt, vo) = q, k, v
    (dqt, dqo), (dkt, dko), (dvt, dvo) = dq, dk, dv
    delta = torch.empty((B, H, Lq), dtype=torch.float32, device=qt.device)
    with torch.cuda.device(qt.device):
        _lib.check(_lib.lib().awt_op_attention_small_backward(
            _ctx(qt), qt.data_ptr() + 4 * qo, ldq, kt.data_ptr() + 4 * ko, ldk, vt.data_ptr() + 4 * vo, ldv, _lib.ptr(o), _lib.ptr(dout), H * 64, _lib.ptr(lse),
            _lib.ptr(delta), dqt.data_ptr() + 4 * dqo, dkt.data_ptr() + 4 * dko, dvt.data_ptr() + 4 * dvo, B, H, Lq, Sk, int(causal), int(causal_off),
            _lib.stream_handle()))


def cross_entropy(logits, labels, vocab):
    """(loss scalar tensor, dlogits [M, ld]) of CrossEntropyLoss(ignore_index=-100) over the first `vocab` columns."""
    M, ld = logits.shape
    flat = labels.reshape(-1)
    bad = (flat != -100) & ((flat < 0) | (flat >= vocab))              # torch.nn.CrossEntropyLoss raises for these; so do we (only -100 is ignored)
    if bool(bad.any()):
        raise IndexError(f"Target {int(flat[bad][0])} is out of bounds.")
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    scratch = torch.empty(M + 1, dtype=torch.float32, device=logits.device)
    lab = labels.reshape(-1).to(device=logits.device, dtype=torch.int64).contiguous()
    with torch.cuda.device(logits.device):
        _lib.check(_lib.lib().awt_op_cross_entropy(_ctx(logits), _lib.ptr(logits), _lib.ptr(lab), M, vocab, ld, _lib.ptr(loss), _lib.ptr(dlogits),
                                                   _lib.ptr(scratch), _lib.stream_handle()))
    return loss, dlogits


class _Leaf(nn.Module):
    pass


class NativeWhisperDecoder(nn.Module):
    """Pre-LN Whisper decoder with tied output projection on libawt.  HF parameter names; all parameters frozen."""

    def __init__(self, d: int, layers: int, heads: int, ffn: int, vocab: int = 51865, max_target_positions: int = 448, precision: str = "bf16x3"):
        super().__init__()
        if d != heads * 64:
            raise ValueError("the native attention kernels are specialised for head_dim 64 (every Whisper size)")
        self.d, self.n_layers, self.heads, self.ffn, self.vocab, self.precision = d, layers, heads, ffn, vocab, precision

        def P(*shape):
            return nn.Parameter(torch.zeros(*shape), requires_grad=False)

        self.embed_tokens = _Leaf(); self.embed_tokens.weight = P(vocab, d)
        self.embed_positions = _Leaf(); self.embed_positions.weight = P(max_target_positions, d)
        self.layers = nn.ModuleList()
        for _ in range(layers):
            lay = _Leaf()
            for att in ("self_attn", "encoder_attn"):
                a = _Leaf()
                for proj, has_bias in (("q_proj", True), ("k_proj", False), ("v_proj", True), ("out_proj", True)):
                    p = _Leaf(); p.weight = P(d, d)
                    if has_bias:
                        p.bias = P(d)
                    setattr(a, proj, p)
                setattr(lay, att, a)
                ln = _Leaf(); ln.weight = P(d); ln.bias = P(d)
                setattr(lay, att + "_layer_norm", ln)
            lay.fc1 = _Leaf(); lay.fc1.weight = P(ffn, d); lay.fc1.bias = P(ffn)
            lay.fc2 = _Leaf(); lay.fc2.weight = P(d, ffn); lay.fc2.bias = P(d)
            lay.final_layer_norm = _Leaf(); lay.final_layer_norm.weight = P(d); lay.final_layer_norm.bias = P(d)
            self.layers.append(lay)
        self.layer_norm = _Leaf(); self.layer_norm.weight = P(d); self.layer_norm.bias = P(d)
        self._packed: Optional[Dict[str, object]] = None
        self._versions: Optional[Tuple[int, ...]] = None

    # ------------------------------------------------------------------------------------------------ packing
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        res = super().load_state_dict(state_dict, strict=strict, **kw)
        self._packed = None
        return res

    def _apply(self, fn, *a, **kw):
        self._packed = None
        return super()._apply(fn, *a, **kw)

    def packed(self) -> Dict[str, object]:
        """Frozen weights as `awt_weight` handles, rebuilt when a parameter changed: per layer the fused self-attention q|k|v
        projection, the two output projections, the cross-attention query projection, fc1, fc2; the fused cross-attention k|v
        projection of ALL layers (one GEMM over the encoder output) and the tied vocabulary projection."""
        vers = tuple(p._version for p in self.parameters())
        if self._packed is not None and self._versions == vers:
            return self._packed
        pk: Dict[str, object] = {"layers": []}
        prec = self.precision
        zeros = torch.zeros(self.d, device=self.embed_tokens.weight.device)
        for lay in self.layers:
            sa, ca = lay.self_attn, lay.encoder_attn
            pk["layers"].append({
                "qkv": PackedLinear(torch.cat([sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight]), torch.cat([sa.q_proj.bias, zeros, sa.v_proj.bias]), prec),
                "so": PackedLinear(sa.out_proj.weight, sa.out_proj.bias, prec),
                "cq": PackedLinear(ca.q_proj.weight, ca.q_proj.bias, prec),
                "co": PackedLinear(ca.out_proj.weight, ca.out_proj.bias, prec),
                "fc1": PackedLinear(lay.fc1.weight, lay.fc1.bias, prec),
                "fc2": PackedLinear(lay.fc2.weight, lay.fc2.bias, prec)})
        att = [l.encoder_attn for l in self.layers]
        pk["ckv"] = PackedLinear(torch.cat([w for a in att for w in (a.k_proj.weight, a.v_proj.weight)]),
                                 torch.cat([b for a in att for b in (zeros, a.v_proj.bias)]), prec)
        pk["vocab"] = PackedLinear(self.embed_tokens.weight, None, prec)
        self._packed, self._versions = pk, vers
        return pk

    # ------------------------------------------------------------------------------------------------ forward pieces
    def cross_kv(self, enc: torch.Tensor, precision: Optional[str] = None) -> torch.Tensor:
        """[B * S, 2 * layers * d]: cross-attention keys (even d-wide blocks) and values (odd blocks) of every layer."""
        B, S, d = enc.shape
        return self.packed()["ckv"].forward(enc.reshape(B * S, d).float().contiguous())

    def _embed(self, ids: torch.Tensor, position_offset: int) -> torch.Tensor:
        B, L = ids.shape
        ids = ids.to(device=self.embed_tokens.weight.device, dtype=torch.int64).contiguous()
        x = torch.empty((B * L, self.d), dtype=torch.float32, device=ids.device)
        with torch.cuda.device(ids.device):
            _lib.check(_lib.lib().awt_op_embed(_lib.ctx(ids.device), _lib.ptr(ids), _lib.ptr(self.embed_tokens.weight), _lib.ptr(self.embed_positions.weight),
                                               _lib.ptr(x), B * L, L, self.d, int(position_offset), self.vocab, _lib.stream_handle()))
        return x

    def _run(self, ids: torch.Tensor, kv: torch.Tensor, S: int, save: Optional[list], caches=None, position_offset: int = 0) -> torch.Tensor:
        """Token ids [B, L] + cross keys / values -> final-LayerNorm input x [B * L, d] (saving what the backward needs in `save`)."""
        pk = self.packed()
        B, L = ids.shape
        d, H, nl = self.d, self.heads, self.n_layers
        x = self._embed(ids, position_offset)
        for i, (lay, p) in enumerate(zip(self.layers, pk["layers"])):
            h = layernorm(x, lay.self_attn_layer_norm.weight, lay.self_attn_layer_norm.bias)
            qkv = p["qkv"].forward(h)                                                   # [B L, 3 d]
            if caches is None:
                a, lse = attention_small((qkv, 0), 3 * d, (qkv, d), 3 * d, (qkv, 2 * d), 3 * d, B, H, L, L, True, 0, save is not None)
            else:                                                                       # incremental decoding: keys / values of all positions so far
                c = caches[i]
                kvn = qkv.view(B, L, 3 * d)[:, :, d:]
                c["kv"] = kvn.contiguous() if "kv" not in c else torch.cat([c["kv"], kvn], dim=1)
                T = c["kv"].shape[1]
                a, lse = attention_small((qkv, 0), 3 * d, (c["kv"], 0), 2 * d, (c["kv"], d), 2 * d, B, H, L, T, True, T - L, False)
            x1 = p["so"].forward(a, resid=x)
            h2 = layernorm(x1, lay.encoder_attn_layer_norm.weight, lay.encoder_attn_layer_norm.bias)
            q = p["cq"].forward(h2)
            a2, lse2 = attention_small((q, 0), d, (kv, 2 * i * d), 2 * nl * d, (kv, (2 * i + 1) * d), 2 * nl * d, B, H, L, S, False, 0, save is not None)
            x2 = p["co"].forward(a2, resid=x1)
            h3 = layernorm(x2, lay.final_layer_norm.weight, lay.final_layer_norm.bias)
            f = p["fc1"].forward(h3)
            x3 = p["fc2"].forward(gelu(f), resid=x2)
            if save is not None:
                save.append((x, qkv, a, lse, x1, q, a2, lse2, x2, f))
            x = x3
        return x

    def forward(self, input_ids: torch.Tensor, encoder_hidden_states: torch.Tensor, native_precision: Optional[str] = None, cross=None, caches=None,
                position_offset: int = 0) -> torch.Tensor:
        """Logits [B, L, vocab] (no autograd graph: training goes through `loss`).  `cross`: the tensor `cross_kv` returned."""
        with torch.no_grad():
            B, L = input_ids.shape
            kv = cross if cross is not None else self.cross_kv(encoder_hidden_states)
            x = self._run(input_ids, kv, encoder_hidden_states.shape[1], None, caches, position_offset)
            xf = layernorm(x, self.layer_norm.weight, self.layer_norm.bias)
            logits = self.packed()["vocab"].forward(xf)
            return logits.view(B, L, -1)[:, :, : self.vocab]

    def loss(self, decoder_input_ids: torch.Tensor, labels: torch.Tensor, encoder_hidden_states: torch.Tensor):
        """(loss, logits [B, L, vocab]): differentiable w.r.t. `encoder_hidden_states` only (the decoder is frozen)."""
        holder: List[torch.Tensor] = []
        loss = _DecoderLoss.apply(encoder_hidden_states, self, decoder_input_ids, labels, holder)
        return loss, holder[0]


class _DecoderLoss(torch.autograd.Function):
    """decoder + tied projection + cross-entropy as ONE autograd node with a hand-written native backward."""

    @staticmethod
    def forward(ctx, enc, dec: NativeWhisperDecoder, ids, labels, holder):
        B, S, d = enc.shape
        L = ids.shape[1]
        kv = dec.cross_kv(enc)
        save: list = []
        x = dec._run(ids, kv, S, save)
        xf = layernorm(x, dec.layer_norm.weight, dec.layer_norm.bias)
        logits = dec.packed()["vocab"].forward(xf)                                          # [B L, Np]
        loss, dlogits = cross_entropy(logits, labels, dec.vocab)
        holder.append(logits.view(B, L, -1)[:, :, : dec.vocab])
        ctx.dec, ctx.saved, ctx.kv, ctx.x_last, ctx.dlogits, ctx.shape = dec, save, kv, x, dlogits, (B, S, L)
        return loss

    @staticmethod
    def backward(ctx, g):
        dec, save, kv, (B, S, L) = ctx.dec, ctx.saved, ctx.kv, ctx.shape
        pk = dec.packed()
        d, H, nl = dec.d, dec.heads, dec.n_layers
        dlogits = ctx.dlogits * g                                                           # upstream scalar (1 / micro-batches, ...)
        dx = layernorm_backward(pk["vocab"].backward_input(dlogits), ctx.x_last, dec.layer_norm.weight)
        dkv = torch.empty_like(kv)                                                          # every layer writes its own two d-wide blocks
        for i in range(nl - 1, -1, -1):
            lay, p = dec.layers[i], pk["layers"][i]
            x0, qkv, a, lse, x1, q, a2, lse2, x2, f = save[i]
            # MLP: x3 = x2 + fc2(gelu(fc1(LN3(x2))))
            df = gelu_backward(f, p["fc2"].backward_input(dx))
            dx = layernorm_backward(p["fc1"].backward_input(df), x2, lay.final_layer_norm.weight, dres=dx)
            # cross-attention: x2 = x1 + out(attn(q(LN2(x1)), K_i, V_i))
            da2 = p["co"].backward_input(dx)
            dq = torch.empty_like(q)
            attention_small_backward((q, 0), d, (kv, 2 * i * d), 2 * nl * d, (kv, (2 * i + 1) * d), 2 * nl * d, a2, da2, lse2,
                                     (dq, 0), (dkv, 2 * i * d), (dkv, (2 * i + 1) * d), B, H, L, S, False, 0)
            dx = layernorm_backward(p["cq"].backward_input(dq), x1, lay.encoder_attn_layer_norm.weight, dres=dx)
            # self-attention: x1 = x0 + out(attn(qkv(LN1(x0))))
            da = p["so"].backward_input(dx)
            dqkv = torch.empty_like(qkv)
            attention_small_backward((qkv, 0), 3 * d, (qkv, d), 3 * d, (qkv, 2 * d), 3 * d, a, da, lse,
                                     (dqkv, 0), (dqkv, d), (dqkv, 2 * d), B, H, L, L, True, 0)
            dx = layernorm_backward(p["qkv"].backward_input(dqkv), x0, lay.self_attn_layer_norm.weight, dres=dx)
        ctx.saved = ctx.kv = ctx.dlogits = None
        d_enc = pk["ckv"].backward_input(dkv)                                               # [B S, d]
        return d_enc.view(B, S, d), None, None, None, None
