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

Cross-attention in the training step runs in its ABSORBED form (`_AbsorbedCross`, csrc/bmm.hip): with L ~ 12 label rows per clip the key /
value projections of the B x 1500 encoder rows (2 x 768 x 768 MACs per row and decoder layer, forward and backward, plus a 7 GB key / value
tensor and its gradient at B = 64) are re-associated onto the query side -- scores = (q_h W_k,h) enc^T, context = (P_h enc) W_v,h^T + b_v --
so that each layer costs two [H L, d] x [d, S] products per clip against the encoder states themselves (packed once per step as the GEMM's B
operand and shared by all layers, forward and backward).  Exact re-association: results differ from the key / value form by rounding only.
`generate` / `forward` (many decoding steps against the same clip) keep the projected keys / values (`cross_kv`).

Parameter names are HF's (`WhisperDecoder.state_dict()` loads unchanged); `forward` has `finetune.WhisperDecoder`'s signature, so
`greedy_decode` / `generate` run on it as well (incremental decoding with a self-attention cache).

LoRA on the decoder (scope row f1, second half; build-defined like the encoder's adapters: y = x W^T + b + (alpha / r) (x A^T) B^T): with
`lora=LoraSpec(targets=("q_proj", "v_proj"))` the self-attention and the cross-attention query / value projections of every layer get
trainable `lora_A` [r, d] / `lora_B` [d, r] (B = 0 at init).  The base stays frozen and packed; the adapter terms and their gradients are
small GEMMs on the same MFMA kernel (`awt_op_linear`): du = dy B, dA = (alpha / r) du^T x, dB = (alpha / r) dy^T u, dx += (alpha / r) du A.
The cross-attention VALUE adapters act on the encoder states (B x 1500 rows): all layers' u = enc A^T come from one GEMM, and their share
of d(loss) / d(encoder states) from one more.
"""
from __future__ import annotations

import ctypes as C
import os
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
VALIDATE_LABELS = os.environ.get("AWT_VALIDATE_LABELS", "0") == "1"   # cross_entropy: range-check device labels on the host (a stream synchronisation per call)

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


def attention_small(q, ldq, k, ldk, v, ldv, B, H, Lq, Sk, causal, causal_off, want_lse=True, drop_p=0.0, seed=0):
    """q / k / v: (tensor, column offset in elements) pairs over row-major fp32 matrices; returns o [B * Lq, H * 64] and lse [B, H, Lq].
    drop_p > 0: attention-probability dropout with the seeded in-kernel mask (include/awt.h awt_op_attention_small_dropout)."""
    (qt, qo), (kt, ko), (vt, vo) = q, k, v
    o = torch.empty((B * Lq, H * 64), dtype=torch.float32, device=qt.device)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=qt.device) if want_lse else None
    with torch.cuda.device(qt.device):
        _lib.check(_lib.lib().awt_op_attention_small_dropout(_ctx(qt), qt.data_ptr() + 4 * qo, ldq, kt.data_ptr() + 4 * ko, ldk, vt.data_ptr() + 4 * vo, ldv,
                                                             _lib.ptr(o), H * 64, _lib.ptr(lse), B, H, Lq, Sk, int(causal), int(causal_off), float(drop_p), int(seed),
                                                             _lib.stream_handle()))
    return o, lse


def attention_small_backward(q, ldq, k, ldk, v, ldv, o, dout, lse, dq, dk, dv, B, H, Lq, Sk, causal, causal_off, drop_p=0.0, seed=0):
    """Writes dq / dk / dv at (tensor, column offset) destinations laid out like q / k / v."""
    (qt, qo), (kt, ko), (vt, vo) = q, k, v
    (dqt, dqo), (dkt, dko), (dvt, dvo) = dq, dk, dv
    delta = torch.empty((B, H, Lq), dtype=torch.float32, device=qt.device)
    with torch.cuda.device(qt.device):
        _lib.check(_lib.lib().awt_op_attention_small_backward_dropout(
            _ctx(qt), qt.data_ptr() + 4 * qo, ldq, kt.data_ptr() + 4 * ko, ldk, vt.data_ptr() + 4 * vo, ldv, _lib.ptr(o), _lib.ptr(dout), H * 64, _lib.ptr(lse),
            _lib.ptr(delta), dqt.data_ptr() + 4 * dqo, dkt.data_ptr() + 4 * dko, dvt.data_ptr() + 4 * dvo, B, H, Lq, Sk, int(causal), int(causal_off),
            float(drop_p), int(seed), _lib.stream_handle()))


def gemm(x: torch.Tensor, w: torch.Tensor, precision: str = "bf16x3") -> torch.Tensor:
    """x [M, K] @ w [N, K]^T on libawt's MFMA GEMM (`awt_op_linear`): K zero-padded to a multiple of 64, N to a multiple of 128."""
    from . import ops
    M, K = x.shape
    N = w.shape[0]
    kp, np_ = (K + 63) // 64 * 64, (N + 127) // 128 * 128
    if kp != K:
        x, w = torch.nn.functional.pad(x, (0, kp - K)), torch.nn.functional.pad(w, (0, kp - K))
    if np_ != N:
        w = torch.nn.functional.pad(w, (0, 0, 0, np_ - N))
    return ops.linear(x.contiguous(), w.contiguous(), None, precision)[:, :N]


class PackedBatch:
    """B operands of `bmm`, packed once into the GEMM's fragment-major weight planes: fp32 [batch, N, K] (contiguous; `awt_bmm_pack`), or with
    `kmajor=((t1, off1), (t2, off2) | None, K1, ld, stride, batch, N, K)` the operand given transposed in memory, B_z[n, k] =
    (k < K1 ? t1 : t2)[off + z stride + (k | k - K1) ld + n] (`awt_bmm_pack_kmajor`)."""

    def __init__(self, b: Optional[torch.Tensor] = None, kmajor=None):
        L = _lib.lib()
        if kmajor is None:
            b = b.detach().float().contiguous()
            self.batch, self.N, self.K = b.shape
            dev = b.device
        else:
            (t1, o1), second, K1, ld, stride, self.batch, self.N, self.K = kmajor
            dev = t1.device
        nbytes = int(L.awt_bmm_packed_bytes(self.batch, self.N, self.K))
        self.buf = _lib.workspace(nbytes, dev)
        with torch.cuda.device(dev):
            if kmajor is None:
                _lib.check(L.awt_bmm_pack(_lib.ctx(dev), _lib.ptr(b), self.K, self.N * self.K, self.batch, self.N, self.K, _lib.ptr(self.buf), self.buf.numel(),
                                          _lib.stream_handle()))
            else:
                p2 = None if second is None else second[0].data_ptr() + 4 * second[1]
                _lib.check(L.awt_bmm_pack_kmajor(_lib.ctx(dev), t1.data_ptr() + 4 * o1, p2, K1, ld, stride, self.batch, self.N, self.K, _lib.ptr(self.buf),
                                                 self.buf.numel(), _lib.stream_handle()))


def bmm(a, pb: PackedBatch, M: int, out, resid=None, a_kmajor=None) -> None:
    """out_z [M, N] = a_z [M, K] B_z^T (+ resid_z) for z < pb.batch.  `a`, `out`, `resid`: (tensor, element offset, row pitch, matrix stride) views
    of row-major fp32 storage (`resid` shares the output's pitch / stride and may be the output itself).  `a_kmajor=((t1, off1), (t2, off2) | None,
    K1, ld, stride)` instead of `a`: the left operand given transposed in memory, optionally as two matrices stacked along K (as PackedBatch)."""
    ot, oo, ldo, so = out
    dev = ot.device
    L = _lib.lib()
    ws = _lib.workspace(L.awt_bmm_workspace_bytes(pb.batch, M, pb.K), dev)
    rp = None
    if resid is not None:
        rt, ro, ldr, sr = resid
        if (ldr, sr) != (ldo, so):
            raise ValueError("bmm: the residual must have the output's pitch and stride")
        rp = rt.data_ptr() + 4 * ro
    with torch.cuda.device(dev):
        if a_kmajor is None:
            at, ao, lda, sa = a
            _lib.check(L.awt_bmm(_lib.ctx(dev), at.data_ptr() + 4 * ao, lda, sa, _lib.ptr(pb.buf), rp, ot.data_ptr() + 4 * oo, ldo, so, pb.batch, M, pb.N, pb.K,
                                 _lib.ptr(ws), ws.numel(), _lib.stream_handle()))
        else:
            (t1, o1), second, K1, lda, sa = a_kmajor
            p2 = None if second is None else second[0].data_ptr() + 4 * second[1]
            _lib.check(L.awt_bmm_kmajor(_lib.ctx(dev), t1.data_ptr() + 4 * o1, p2, K1, lda, sa, _lib.ptr(pb.buf), rp, ot.data_ptr() + 4 * oo, ldo, so, pb.batch,
                                        M, pb.N, pb.K, _lib.ptr(ws), ws.numel(), _lib.stream_handle()))


def softmax_rows(s: torch.Tensor, cols: int, scale: float) -> torch.Tensor:
    """In place: rows of the 2-D view [rows, ld] -> softmax(scale * row[:cols])."""
    rows, ld = s.shape
    with torch.cuda.device(s.device):
        _lib.check(_lib.lib().awt_op_softmax_rows(_ctx(s), _lib.ptr(s), _lib.ptr(s), rows, cols, ld, scale, _lib.stream_handle()))
    return s


def softmax_rows_backward(p: torch.Tensor, dp: torch.Tensor, cols: int, scale: float) -> torch.Tensor:
    """In place on dp: scale * p * (dp - rowsum(p dp))."""
    rows, ld = p.shape
    with torch.cuda.device(p.device):
        _lib.check(_lib.lib().awt_op_softmax_rows_backward(_ctx(p), _lib.ptr(p), _lib.ptr(dp), _lib.ptr(dp), rows, cols, ld, scale, _lib.stream_handle()))
    return dp


class _AbsorbedCross:
    """One step's cross-attention over the encoder states `enc` [B, S, d] with the key / value projections moved to the query side (module
    docstring).  Rows of the per-clip matrices are ordered r = l * H + h, so that [B L, H, d] storage is at once `H` strided per-head matrices
    (pitch H d, stride d) and `B` per-clip matrices [L H, d].  HF:modeling_whisper.py:284-356."""

    ROW_LIMIT = 512          # H L above which the projected keys / values (cross_kv) are the cheaper form: 6 H L > 4 d

    def __init__(self, dec: "NativeWhisperDecoder", enc: torch.Tensor):
        self.dec = dec
        self.B, self.S, self.d = enc.shape
        self.Sp = (self.S + 3) // 4 * 4
        self.enc_p = PackedBatch(enc)                                           # B operand [S, d] of scores and dP
        self.encT_p = PackedBatch(enc.transpose(1, 2))                          # B operand [d, S] of the context and d(q~)
        self.eff: Dict[int, dict] = {}

    def _heads(self, i: int) -> dict:
        """Per-head blocks of layer i's key / value projections as batched B operands; with a value adapter the merged weight
        W_v + (alpha / r) B A of this step (the adapter's gradient comes back through d(merged weight), `backward`)."""
        dec, H, d = self.dec, self.dec.heads, self.d
        pk = dec.packed()["cross_heads"][i]
        ab = dec._adapter(i, "encoder_attn", "v_proj")
        if ab is None:
            return pk
        if i not in self.eff:
            A, Bm = ab
            wv = dec.layers[i].encoder_attn.v_proj.weight.detach() + gemm(Bm.detach(), A.detach().t().contiguous(), dec.precision) * dec.lora.scale
            self.eff[i] = dict(pk, v=PackedBatch(wv.view(H, 64, d)), vT=PackedBatch(wv.view(H, 64, d).transpose(1, 2)))
        return self.eff[i]

    def attend(self, i: int, q: torch.Tensor, L: int):
        """q [B L, d] (bias and adapter applied, unscaled) -> (attention output [B L, d] before out_proj, what `backward` needs)."""
        B, S, Sp, d, H = self.B, self.S, self.Sp, self.d, self.dec.heads
        R, M = L * H, B * L
        hw = self._heads(i)
        dev = q.device
        qt = torch.empty((M, H, d), dtype=torch.float32, device=dev)
        bmm((q, 0, d, 64), hw["kT"], M, (qt, 0, H * d, d))                                      # q~_h = q_h W_k,h
        P = torch.empty((B * R, Sp), dtype=torch.float32, device=dev)
        bmm((qt, 0, d, R * d), self.enc_p, R, (P, 0, Sp, R * Sp))                                # scores = q~ enc^T
        softmax_rows(P, S, 0.125)
        c = torch.empty((M, H, d), dtype=torch.float32, device=dev)
        bmm((P, 0, Sp, R * Sp), self.encT_p, R, (c, 0, d, R * d))                                # context = P enc
        a2 = torch.empty((M, d), dtype=torch.float32, device=dev)
        bmm((c, 0, H * d, d), hw["v"], M, (a2, 0, d, 64))                                        # (P enc) W_v,h^T
        a2 += self.dec.layers[i].encoder_attn.v_proj.bias                                       # rows of P sum to one
        return a2, (qt, P, c)

    def backward(self, i: int, saved, da2: torch.Tensor, d_enc: torch.Tensor, L: int, grads: dict) -> torch.Tensor:
        """Adds this layer's share of d(loss) / d(encoder states) to `d_enc` [B, S, d]; returns dq [B L, d]."""
        B, S, Sp, d, H = self.B, self.S, self.Sp, self.d, self.dec.heads
        R, M = L * H, B * L
        qt, P, c = saved
        hw = self._heads(i)
        dev = da2.device
        da2 = da2.contiguous()
        dc = torch.empty((M, H, d), dtype=torch.float32, device=dev)
        bmm((da2, 0, d, 64), hw["vT"], M, (dc, 0, H * d, d))                                     # d(context)_h = da2_h W_v,h
        dS = torch.empty((B * R, Sp), dtype=torch.float32, device=dev)
        bmm((dc, 0, d, R * d), self.enc_p, R, (dS, 0, Sp, R * Sp))                               # dP = d(context) enc^T
        softmax_rows_backward(P, dS, S, 0.125)
        dqt = torch.empty((M, H, d), dtype=torch.float32, device=dev)
        bmm((dS, 0, Sp, R * Sp), self.encT_p, R, (dqt, 0, d, R * d))                             # d(q~) = dS enc
        # d(enc)_b += P_b^T d(context)_b + dS_b^T q~_b : one [S, 2 R] x [2 R, d] product per clip
        # (all four matrices are read where they lie: both operands are K-major, two blocks stacked along K)
        rhs = PackedBatch(kmajor=((dc, 0), (qt, 0), R, d, R * d, B, d, 2 * R))                   # B operand [d, 2 R] = [d(context)_b ; q~_b]^T
        bmm(None, rhs, S, (d_enc, 0, d, S * d), resid=(d_enc, 0, d, S * d), a_kmajor=((P, 0), (dS, 0), R, Sp, R * Sp))
        dq = torch.empty((M, d), dtype=torch.float32, device=dev)
        bmm((dqt, 0, H * d, d), hw["k"], M, (dq, 0, d, 64))                                      # dq_h = d(q~)_h W_k,h^T
        ab = self.dec._adapter(i, "encoder_attn", "v_proj")
        if ab is not None:                                                                       # d(merged W_v)_h = da2_h^T context_h, then through W_v + s B A
            A, Bm = ab
            dw = torch.empty((d, d), dtype=torch.float32, device=dev)
            bmm(None, PackedBatch(kmajor=((c, 0), None, M, H * d, d, H, d, M)), 64, (dw, 0, d, 64 * d), a_kmajor=((da2, 0), None, M, d, 64))
            scale, prec = self.dec.lora.scale, self.dec.precision
            grads[id(Bm)] = gemm(dw, A.detach(), prec) * scale                                   # dB [d, r] = s dW A^T
            grads[id(A)] = gemm(Bm.detach().t().contiguous(), dw.t().contiguous(), prec) * scale  # dA [r, d] = s B^T dW
        return dq


def cross_entropy(logits, labels, vocab):
    """(loss scalar tensor, dlogits [M, ld]) of CrossEntropyLoss(ignore_index=-100) over the first `vocab` columns."""
    M, ld = logits.shape
    flat = labels.reshape(-1)
    if VALIDATE_LABELS or not flat.is_cuda:
        # torch.nn.CrossEntropyLoss raises for targets outside [0, vocab) other than -100; so do we -- but on device labels that check is a host
        # synchronisation in front of every loss launch, so it runs only for host labels (free) or when AWT_VALIDATE_LABELS=1 asks for it:
        # the kernel itself poisons the loss with NaN for an out-of-range device target (tests/test_gpu_native_decoder.py), which a training loop sees at once
        bad = (flat != -100) & ((flat < 0) | (flat >= vocab))
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

    LORA_TARGETS = ("q_proj", "v_proj")

    def __init__(self, d: int, layers: int, heads: int, ffn: int, vocab: int = 51865, max_target_positions: int = 448, precision: str = "bf16x3",
                 lora=None, lora_seed: int = 0):
        super().__init__()
        if d != heads * 64:
            raise ValueError("the native attention kernels are specialised for head_dim 64 (every Whisper size)")
        self.d, self.n_layers, self.heads, self.ffn, self.vocab, self.precision = d, layers, heads, ffn, vocab, precision
        if lora is not None and not set(lora.targets) <= set(self.LORA_TARGETS):
            raise ValueError(f"decoder adapters are built for {self.LORA_TARGETS} of self_attn and encoder_attn, got {lora.targets}")
        self.lora = lora

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
        if lora is not None:
            import math
            from .weights import unit_variates
            for i, lay in enumerate(self.layers):
                for att in ("self_attn", "encoder_attn"):
                    for proj in lora.targets:
                        leaf = getattr(getattr(lay, att), proj)
                        a = unit_variates(f"decoder.layers.{i}.{att}.{proj}.lora_A", lora.r * d, lora_seed) / math.sqrt(d)
                        leaf.lora_A = nn.Parameter(torch.from_numpy(a.astype("float32").reshape(lora.r, d)), requires_grad=True)
                        leaf.lora_B = nn.Parameter(torch.zeros(d, lora.r), requires_grad=True)          # B = 0: the adapter starts inert
        self._packed: Optional[Dict[str, object]] = None
        self._versions: Optional[Tuple[int, ...]] = None
        import os
        self.cross_mode = os.environ.get("AWT_DECODER_CROSS", "auto")          # "auto" | "kv" | "absorbed" (A/B runs: tools, bench)
        if self.cross_mode not in ("auto", "kv", "absorbed"):
            raise ValueError("AWT_DECODER_CROSS must be auto, kv or absorbed")

    def absorbed_cross(self, L: int, S: int) -> bool:
        """Whether the training step's cross-attention takes the absorbed form (`cross_mode`: "auto" = by the row count H L, "kv", "absorbed")."""
        if self.cross_mode == "auto":
            return self.heads * L <= _AbsorbedCross.ROW_LIMIT and S % 4 == 0 and S <= 4096
        if self.cross_mode == "absorbed" and (S % 4 != 0 or S > 4096):
            raise ValueError("absorbed cross-attention needs a multiple of 4 and at most 4096 encoder positions")
        return self.cross_mode == "absorbed"

    # ------------------------------------------------------------------------------------------------ adapters
    def lora_parameters(self) -> List[nn.Parameter]:
        """Adapter parameters in a fixed order: per layer, self_attn then encoder_attn, per target: lora_A then lora_B."""
        out: List[nn.Parameter] = []
        if self.lora is not None:
            for lay in self.layers:
                for att in ("self_attn", "encoder_attn"):
                    for proj in self.lora.targets:
                        leaf = getattr(getattr(lay, att), proj)
                        out += [leaf.lora_A, leaf.lora_B]
        return out

    def _adapter(self, i: int, att: str, proj: str):
        if self.lora is None or proj not in self.lora.targets:
            return None
        leaf = getattr(getattr(self.layers[i], att), proj)
        return leaf.lora_A, leaf.lora_B

    def _lora_term(self, x: torch.Tensor, ab) -> Tuple[torch.Tensor, torch.Tensor]:
        """(u = x A^T [M, r], (alpha / r) u B^T [M, d]) on the native GEMM."""
        A, Bm = ab
        u = gemm(x, A.detach(), self.precision)
        return u, gemm(u, Bm.detach(), self.precision) * self.lora.scale

    def _cross_v_adapters(self, enc2d: torch.Tensor, kv: torch.Tensor):
        """Adds (alpha / r) (enc A_i^T) B_i^T to layer i's VALUE block of the fused cross-attention projection; returns u of all layers
        [B * S, layers * r] (one GEMM over the encoder states), or None without value adapters."""
        if self.lora is None or "v_proj" not in self.lora.targets:
            return None
        r, d, nl = self.lora.r, self.d, self.n_layers
        a_all = torch.cat([self.layers[i].encoder_attn.v_proj.lora_A.detach() for i in range(nl)], dim=0)        # [layers * r, d]
        u_all = gemm(enc2d, a_all, self.precision)
        blocks = kv.view(kv.shape[0], 2 * nl, d)
        for i in range(nl):
            blocks[:, 2 * i + 1] += gemm(u_all[:, i * r:(i + 1) * r], self.layers[i].encoder_attn.v_proj.lora_B.detach(), self.precision) * self.lora.scale
        return u_all

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
        vers = tuple((id(p), p._version) for n, p in self.named_parameters() if "lora_" not in n)     # identity too: a replaced Parameter object starts at version 0 again
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
        H, d = self.heads, self.d
        pk["cross_heads"] = []                                                     # per-head blocks of W_k / W_v as batched B operands (_AbsorbedCross)
        for a in att:
            wk, wv = a.k_proj.weight.detach().view(H, 64, d), a.v_proj.weight.detach().view(H, 64, d)
            pk["cross_heads"].append({"k": PackedBatch(wk), "kT": PackedBatch(wk.transpose(1, 2)), "v": PackedBatch(wv), "vT": PackedBatch(wv.transpose(1, 2))})
        self._packed, self._versions = pk, vers
        return pk

    # ------------------------------------------------------------------------------------------------ forward pieces
    def cross_kv(self, enc: torch.Tensor, precision: Optional[str] = None) -> torch.Tensor:
        """[B * S, 2 * layers * d]: cross-attention keys (even d-wide blocks) and values (odd blocks) of every layer."""
        B, S, d = enc.shape
        enc2d = enc.reshape(B * S, d).float().contiguous()
        kv = self.packed()["ckv"].forward(enc2d)
        self._cross_v_adapters(enc2d, kv)
        return kv

    def _embed(self, ids: torch.Tensor, position_offset: int) -> torch.Tensor:
        B, L = ids.shape
        ids = ids.to(device=self.embed_tokens.weight.device, dtype=torch.int64).contiguous()
        x = torch.empty((B * L, self.d), dtype=torch.float32, device=ids.device)
        with torch.cuda.device(ids.device):
            _lib.check(_lib.lib().awt_op_embed(_lib.ctx(ids.device), _lib.ptr(ids), _lib.ptr(self.embed_tokens.weight), _lib.ptr(self.embed_positions.weight),
                                               _lib.ptr(x), B * L, L, self.d, int(position_offset), self.vocab, _lib.stream_handle()))
        return x

    def _run(self, ids: torch.Tensor, kv: torch.Tensor, S: int, save: Optional[list], caches=None, position_offset: int = 0) -> torch.Tensor:
        """Token ids [B, L] + cross keys / values (the `cross_kv` tensor, or an `_AbsorbedCross` over the encoder states) -> final-LayerNorm
        input x [B * L, d] (saving what the backward needs in `save`)."""
        pk = self.packed()
        B, L = ids.shape
        d, H, nl = self.d, self.heads, self.n_layers
        x = self._embed(ids, position_offset)
        for i, (lay, p) in enumerate(zip(self.layers, pk["layers"])):
            h = layernorm(x, lay.self_attn_layer_norm.weight, lay.self_attn_layer_norm.bias)
            qkv = p["qkv"].forward(h)                                                   # [B L, 3 d]
            us = {}
            for proj, col in (("q_proj", 0), ("v_proj", 2 * d)):                         # adapters on the self-attention query / value projections
                ab = self._adapter(i, "self_attn", proj)
                if ab is not None:
                    us[proj], delta = self._lora_term(h, ab)
                    qkv[:, col: col + d] += delta
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
            ab = self._adapter(i, "encoder_attn", "q_proj")
            if ab is not None:
                us["cq"], delta = self._lora_term(h2, ab)
                q += delta
            if isinstance(kv, _AbsorbedCross):
                a2, lse2 = kv.attend(i, q, L)                                            # lse2: (q~, P, context) for the backward
            else:
                a2, lse2 = attention_small((q, 0), d, (kv, 2 * i * d), 2 * nl * d, (kv, (2 * i + 1) * d), 2 * nl * d, B, H, L, S, False, 0, save is not None)
            x2 = p["co"].forward(a2, resid=x1)
            h3 = layernorm(x2, lay.final_layer_norm.weight, lay.final_layer_norm.bias)
            f = p["fc1"].forward(h3)
            x3 = p["fc2"].forward(gelu(f), resid=x2)
            if save is not None:
                save.append((x, qkv, a, lse, x1, q, a2, lse2, x2, f, us))
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
        loss = _DecoderLoss.apply(encoder_hidden_states, self, decoder_input_ids, labels, holder, *self.lora_parameters())
        return loss, holder[0]


class _DecoderLoss(torch.autograd.Function):
    """decoder + tied projection + cross-entropy as ONE autograd node with a hand-written native backward: d(loss) / d(encoder states) and,
    with decoder adapters, d(loss) / d(lora_A, lora_B) in `NativeWhisperDecoder.lora_parameters()` order."""

    @staticmethod
    def forward(ctx, enc, dec: NativeWhisperDecoder, ids, labels, holder, *adapters):
        B, S, d = enc.shape
        L = ids.shape[1]
        enc2d = enc.reshape(B * S, d).float().contiguous()
        if dec.absorbed_cross(L, S):
            kv, u_cv = _AbsorbedCross(dec, enc2d.view(B, S, d)), None
        else:
            kv = dec.packed()["ckv"].forward(enc2d)
            u_cv = dec._cross_v_adapters(enc2d, kv)
        save: list = []
        x = dec._run(ids, kv, S, save)
        xf = layernorm(x, dec.layer_norm.weight, dec.layer_norm.bias)
        logits = dec.packed()["vocab"].forward(xf)                                          # [B L, Np]
        loss, dlogits = cross_entropy(logits, labels, dec.vocab)
        holder.append(logits.view(B, L, -1)[:, :, : dec.vocab])
        ctx.dec, ctx.saved, ctx.kv, ctx.x_last, ctx.dlogits, ctx.shape = dec, save, kv, x, dlogits, (B, S, L)
        ctx.enc2d, ctx.u_cv, ctx.n_adapters = (enc2d if u_cv is not None else None), u_cv, len(adapters)
        return loss

    @staticmethod
    def backward(ctx, g):
        dec, save, kv, (B, S, L) = ctx.dec, ctx.saved, ctx.kv, ctx.shape
        pk = dec.packed()
        d, H, nl = dec.d, dec.heads, dec.n_layers
        prec = dec.precision
        scale = dec.lora.scale if dec.lora is not None else 0.0
        grads: Dict[int, torch.Tensor] = {}                                                 # id(parameter) -> gradient

        def adapter_backward(ab, x_in, u, dy):
            """Adapter (A, B) on input x_in with u = x_in A^T: records dA, dB and returns the adapter's share of d(x_in)."""
            A, Bm = ab
            du = gemm(dy, Bm.detach().t().contiguous(), prec)                               # [M, r] = dy B
            grads[id(A)] = gemm(du.t().contiguous(), x_in.t().contiguous(), prec) * scale   # dA [r, d] = (alpha / r) du^T x
            grads[id(Bm)] = gemm(dy.t().contiguous(), u.t().contiguous(), prec) * scale     # dB [d, r] = (alpha / r) dy^T u
            return gemm(du, A.detach().t().contiguous(), prec) * scale                      # [M, d] = (alpha / r) du A

        dlogits = ctx.dlogits * g                                                           # upstream scalar (1 / micro-batches, ...)
        dx = layernorm_backward(pk["vocab"].backward_input(dlogits), ctx.x_last, dec.layer_norm.weight)
        absorbed = isinstance(kv, _AbsorbedCross)
        if absorbed:
            d_enc = torch.zeros((B, S, d), dtype=torch.float32, device=dx.device)           # every layer adds its share (one batched GEMM each)
        else:
            dkv = torch.empty_like(kv)                                                      # every layer writes its own two d-wide blocks
        for i in range(nl - 1, -1, -1):
            lay, p = dec.layers[i], pk["layers"][i]
            x0, qkv, a, lse, x1, q, a2, lse2, x2, f, us = save[i]
            # MLP: x3 = x2 + fc2(gelu(fc1(LN3(x2))))
            df = gelu_backward(f, p["fc2"].backward_input(dx))
            dx = layernorm_backward(p["fc1"].backward_input(df), x2, lay.final_layer_norm.weight, dres=dx)
            # cross-attention: x2 = x1 + out(attn(q(LN2(x1)), K_i, V_i))
            da2 = p["co"].backward_input(dx)
            if absorbed:
                dq = kv.backward(i, lse2, da2, d_enc, L, grads)
            else:
                dq = torch.empty_like(q)
                attention_small_backward((q, 0), d, (kv, 2 * i * d), 2 * nl * d, (kv, (2 * i + 1) * d), 2 * nl * d, a2, da2, lse2,
                                         (dq, 0), (dkv, 2 * i * d), (dkv, (2 * i + 1) * d), B, H, L, S, False, 0)
            dh2 = p["cq"].backward_input(dq)
            ab = dec._adapter(i, "encoder_attn", "q_proj")
            if ab is not None:
                h2 = layernorm(x1, lay.encoder_attn_layer_norm.weight, lay.encoder_attn_layer_norm.bias)      # recomputed: B x L rows
                dh2 = dh2 + adapter_backward(ab, h2, us["cq"], dq)
            dx = layernorm_backward(dh2, x1, lay.encoder_attn_layer_norm.weight, dres=dx)
            # self-attention: x1 = x0 + out(attn(qkv(LN1(x0))))
            da = p["so"].backward_input(dx)
            dqkv = torch.empty_like(qkv)
            attention_small_backward((qkv, 0), 3 * d, (qkv, d), 3 * d, (qkv, 2 * d), 3 * d, a, da, lse,
                                     (dqkv, 0), (dqkv, d), (dqkv, 2 * d), B, H, L, L, True, 0)
            dh = p["qkv"].backward_input(dqkv)
            if us:
                h = None
                for proj, col in (("q_proj", 0), ("v_proj", 2 * d)):
                    ab = dec._adapter(i, "self_attn", proj)
                    if ab is not None:
                        if h is None:
                            h = layernorm(x0, lay.self_attn_layer_norm.weight, lay.self_attn_layer_norm.bias)
                        dh = dh + adapter_backward(ab, h, us[proj], dqkv[:, col: col + d].contiguous())
            dx = layernorm_backward(dh, x0, lay.self_attn_layer_norm.weight, dres=dx)
        ctx.saved = ctx.kv = ctx.dlogits = None
        if not absorbed:
            d_enc = pk["ckv"].backward_input(dkv)                                           # [B S, d]
        if ctx.u_cv is not None:
            # cross-attention value adapters: dV_i = dkv block 2 i + 1 (B x S rows); their share of d(encoder states) is ONE GEMM over all layers
            r = dec.lora.r
            dv = dkv.view(dkv.shape[0], 2 * nl, d)
            du_all = torch.empty_like(ctx.u_cv)
            enc_t = ctx.enc2d.t().contiguous()
            for i in range(nl):
                A, Bm = dec._adapter(i, "encoder_attn", "v_proj")
                dvi = dv[:, 2 * i + 1].contiguous()
                du = gemm(dvi, Bm.detach().t().contiguous(), prec)
                du_all[:, i * r:(i + 1) * r] = du
                grads[id(Bm)] = gemm(dvi.t().contiguous(), ctx.u_cv[:, i * r:(i + 1) * r].t().contiguous(), prec) * scale
                grads[id(A)] = gemm(du.t().contiguous(), enc_t, prec) * scale
            a_all_t = torch.cat([dec._adapter(i, "encoder_attn", "v_proj")[0].detach() for i in range(nl)], dim=0).t().contiguous()   # [d, layers * r]
            d_enc = d_enc + gemm(du_all, a_all_t, prec) * scale
            ctx.enc2d = ctx.u_cv = None
        params = dec.lora_parameters() if ctx.n_adapters else []
        return (d_enc.view(B, S, d), None, None, None, None, *[grads.get(id(p)) for p in params])
