"""`NativeWhisperEncoder`: the reference's encoder call surface over the HIP encoder in libawt.

Reference interface kept:
* `encoder(input_features) -> obj.last_hidden_state` [B, 1500, d]; reachable as `WhisperModel.get_encoder()`
      /root/reference/.charles/music2midi/model.py:33,109-110 ; implicitly /root/reference/AB/fineTune.py:199
* it is an `nn.Module` with `.parameters()`, `.eval()`, `.config.d_model`, `.device` (model.py:37-40,105,120,227)
* `ValueError` when the mel length is not 2 * max_source_positions; `attention_mask` accepted and ignored
      HF:models/whisper/modeling_whisper.py:607-616
* parameters carry the HF state-dict keys (`conv1.weight`, `layers.0.self_attn.q_proj.weight`, ...), so a locally
  present Whisper checkpoint's encoder state dict loads with `load_state_dict`; LoRA adds `<module>.lora_A/B`.

All arithmetic is in libawt (`awt_encoder_forward`); this class only owns the fp32 master parameters and pushes
them to the library when they change.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .weights import EncoderConfig, LoraSpec, encoder_param_shapes, init_encoder_weights, init_lora_weights, lora_param_shapes

# operand precision modes (csrc/common.h PREC_*):
#   bf16    one bf16 MFMA per fragment pair (fast; misses the 1e-3 bound)
#   bf16x3  split-bf16, three bf16 MFMAs (2^-17 per operand; the training path's format)
#   fp16x3  split-fp16, three fp16 MFMAs (2^-23 per operand: within fp32's own noise of the reference, also under 30x outlier gains)
#   f16f8   fp16 main product + the two cross terms on the block-scaled e4m3 MFMA: two MFMA-equivalents (2^-16 per operand)
PRECISIONS = {"bf16": 1, "fp16": 2, "bf16x3": 3, "fp16x3": 4, "f16f8": 5}   # "bf16" / "fp16": one product per fragment pair, measurement modes that miss the 1e-3 bound
DEFAULT_PRECISION = "f16f8"


@dataclass
class BaseModelOutput:
    last_hidden_state: torch.Tensor
    hidden_states: Optional[tuple] = None
    attentions: Optional[tuple] = None

    def __getitem__(self, i):
        return (self.last_hidden_state,)[i]


class _EncoderLoRAFunction(torch.autograd.Function):
    """Native forward (activations kept in a library-owned layout) and native backward to the LoRA parameters.
    The frozen base weights and the input features get no gradient (nothing below the first adapter needs one)."""

    @staticmethod
    def forward(ctx, enc, x, *lora_params):
        L = _lib.lib()
        B, _, T = x.shape
        nbytes = L.awt_encoder_train_workspace_bytes(enc._handle, B)
        saved = _lib.workspace(nbytes, x.device)
        out = torch.empty((B, enc.cfg.max_source_positions, enc.cfg.d_model), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.awt_encoder_forward_train(enc._handle, _lib.ptr(x), B, T, _lib.ptr(out), _lib.ptr(saved), saved.numel(),
                                                   _lib.stream_handle()))
        ctx.enc, ctx.saved, ctx.B = enc, saved, B
        ctx.shapes = [tuple(p.shape) for p in lora_params]
        return out

    @staticmethod
    def backward(ctx, d_out):
        enc, L = ctx.enc, _lib.lib()
        d_out = d_out.to(torch.float32).contiguous()
        n = L.awt_encoder_lora_grad_count(enc._handle)
        if enc.backward_precision == "f16f8":
            # fp16 planes want gradients of order one: the pass carries 2^k x the gradient (largest |d loss / d hidden| -> ~2^6) and the library divides
            # the adapter gradients by 2^k on the way out.  One scalar read-back per step; a zero or non-finite gradient keeps k = 0.
            amax = float(d_out.abs().max())
            k = 0 if not (amax > 0.0 and math.isfinite(amax)) else max(-60, min(60, 6 - math.frexp(amax)[1]))
            _lib.check(L.awt_encoder_set_grad_scale_log2(enc._handle, k))
        if enc._grad_flat is not None:
            # bound gradient buffer (bind_grad_buffer): the library writes / accumulates the adapter gradients straight into
            # the flat buffer the parameters' .grad tensors are views of, and -- when asked -- averages them over the ranks
            # inside the same call; autograd gets no tensors for them
            flags = 0 if (enc._grad_fresh or enc._grad_params[0].grad is None) else _lib.BWD_ACCUMULATE
            if enc.grad_sync and enc._comm is not None:
                flags |= _lib.BWD_ALLREDUCE
            with torch.cuda.device(d_out.device):
                _lib.check(L.awt_encoder_backward_ex(enc._handle, _lib.ptr(d_out), ctx.B, _lib.ptr(ctx.saved), ctx.saved.numel(),
                                                     _lib.ptr(enc._grad_flat), n, flags, _lib.stream_handle()))
            ctx.saved = None
            enc._grad_fresh = False
            for p, v in zip(enc._grad_params, enc._grad_views):
                if p.grad is not v:
                    p.grad = v
            return (None, None, *([None] * len(ctx.shapes)))
        flat = torch.empty(n, dtype=torch.float32, device=d_out.device)
        with torch.cuda.device(d_out.device):
            _lib.check(L.awt_encoder_backward(enc._handle, _lib.ptr(d_out), ctx.B, _lib.ptr(ctx.saved), ctx.saved.numel(), _lib.ptr(flat), n,
                                              _lib.stream_handle()))
        ctx.saved = None
        grads, off = [], 0
        for shp in ctx.shapes:          # library order == lora_param order: per layer, per target (q, k, v): A then B
            k = shp[0] * shp[1]
            grads.append(flat[off: off + k].view(shp))
            off += k
        return (None, None, *grads)


class _Leaf(nn.Module):
    """Parameter holder so that state-dict keys read `<path>.weight` / `.bias` / `.lora_A` / `.lora_B`.  `epoch` is a one-element list
    shared with the owning encoder: replacing or deleting a Parameter object bumps it (NativeWhisperEncoder.sync_weights)."""

    def __init__(self, epoch=None):
        super().__init__()
        object.__setattr__(self, "_awt_epoch", epoch if epoch is not None else [0])

    def register_parameter(self, name, param):
        super().register_parameter(name, param)
        self._awt_epoch[0] += 1

    def __setattr__(self, name, value):
        if isinstance(value, nn.Parameter) or name in self.__dict__.get("_parameters", ()):
            self._awt_epoch[0] += 1
        super().__setattr__(name, value)

    def __delattr__(self, name):
        if name in self.__dict__.get("_parameters", ()):
            self._awt_epoch[0] += 1
        super().__delattr__(name)


def _attach(root: nn.Module, dotted: str, p: nn.Parameter, epoch=None) -> None:
    parts = dotted.split(".")
    mod = root
    for name in parts[:-1]:
        if not hasattr(mod, name):
            mod.add_module(name, _Leaf(epoch))
        mod = getattr(mod, name)
    mod.register_parameter(parts[-1], p)


class NativeWhisperEncoder(nn.Module):
    def __init__(self, cfg: EncoderConfig, precision: Optional[str] = None, lora: Optional[LoraSpec] = None,
                 device: str = "cuda", chunk_clips: int = 0, seed: Optional[int] = 0, init_profile: str = "hf",
                 trainable: bool = False, backward_precision: Optional[str] = None, probe_clips=None):
        super().__init__()
        # probe_clips (precision=None only): the caller's own audio -- 1-D float arrays / tensors at 16 kHz, e.g. a few clips of the data to be encoded --
        # joins the two synthetic clips of the load-time precision probe, so that the measured decision sees what the checkpoint does on real input
        self.probe_clips = [np.asarray(c.detach().cpu() if isinstance(c, torch.Tensor) else c, dtype=np.float32).reshape(-1) for c in (probe_clips or [])]
        # precision=None: training keeps bf16 planes; inference takes the fastest mode that meets the 1e-3 bound for the weights at hand --
        # f16f8, unless the checkpoint has outlier channels (choose_precision), where the 15-bit scheme's relative error turns into
        # absolute errors above the bound and the split-fp16 mode is used instead.  Re-decided when base weights change.
        self._auto_precision = precision is None and not trainable
        if precision is None:
            precision = "bf16x3" if trainable else DEFAULT_PRECISION
        if backward_precision not in (None, precision, "bf16", "f16f8"):
            raise ValueError("backward_precision must be None (= precision), 'bf16' or 'f16f8' (the MLP's backward GEMMs in the f16f8 operand format)")
        if backward_precision == "f16f8" and (precision != "bf16x3" or not trainable or (lora is not None and {"fc1", "fc2"} & set(lora.targets))):
            raise ValueError("backward_precision='f16f8' needs trainable=True, precision='bf16x3' and no adapters on fc1 / fc2")
        self.backward_precision = backward_precision
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        if cfg.head_dim != 64:
            raise ValueError("the native attention kernel is specialised for head_dim 64 (every Whisper size)")
        if trainable and precision not in ("bf16", "bf16x3"):
            raise ValueError("trainable=True keeps its activations as bf16 planes: precision must be 'bf16x3' or 'bf16'")
        if trainable and lora is None:
            raise ValueError("trainable=True needs LoRA adapters")
        self.cfg = cfg
        self.precision = precision
        self.lora = lora
        self.trainable = trainable
        self.config = SimpleNamespace(d_model=cfg.d_model, encoder_layers=cfg.layers, encoder_attention_heads=cfg.heads,
                                      encoder_ffn_dim=cfg.ffn, num_mel_bins=cfg.n_mels,
                                      max_source_positions=cfg.max_source_positions)
        dev = torch.device(device)
        self._epoch = [0]            # bumped by every leaf whose Parameter objects change (see _Leaf)
        base = init_encoder_weights(cfg, seed or 0, init_profile) if seed is not None else None
        for name, shape in encoder_param_shapes(cfg):
            t = torch.from_numpy(base[name]) if base is not None else torch.zeros(shape)
            _attach(self, name, nn.Parameter(t.to(dev), requires_grad=False), self._epoch)
        if lora is not None:
            lw = init_lora_weights(cfg, lora, seed or 0) if seed is not None else None
            for name, shape in lora_param_shapes(cfg, lora):
                t = torch.from_numpy(lw[name]) if lw is not None else torch.zeros(shape)
                _attach(self, name, nn.Parameter(t.to(dev), requires_grad=True), self._epoch)
        self._handle = None
        self._chunk = chunk_clips
        self._synced: Dict[str, int] = {}
        self._param_cache = None
        self._param_ids = None
        self._ws: Optional[torch.Tensor] = None
        # bound gradient buffer (bind_grad_buffer) and the communicator its exchange runs on (set_comm)
        self._grad_flat: Optional[torch.Tensor] = None
        self._grad_params: list = []
        self._grad_views: list = []
        self._grad_fresh = True
        self._comm = None
        self.grad_sync = True       # False: this backward only accumulates (all but the last micro-batch of a step)

    # ------------------------------------------------------------------------------------------------ plumbing
    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def dtype(self) -> torch.dtype:
        return torch.float32

    def get_input_embeddings(self):
        return self.conv1

    # outlier thresholds of the automatic mode: largest LayerNorm gain over the median gain, largest out_proj / fc2 row norm over the median
    AUTO_GAIN_RATIO, AUTO_ROW_RATIO = 8.0, 5.0

    def choose_precision(self) -> str:
        """'f16f8' or 'fp16x3' from the weights alone: LayerNorm gains and out_proj / fc2 rows far above their medians are what turns a
        relative operand error into a large absolute one (tests/test_gpu_encoder.py outlier profile: f16f8 8e-2, fp16x3 7e-4 at outputs of
        66).  One device reduction per tensor, one host sync in total; `self.precision_report` keeps the two ratios."""
        stats = []
        for name, p in self.named_parameters():
            if name.endswith("layer_norm.weight"):
                a = p.detach().abs().float()
                stats.append(torch.stack([a.max() / a.median().clamp_min(1e-12), a.new_zeros(())]))
            elif name.endswith(("out_proj.weight", "fc2.weight")):
                n = p.detach().float().norm(dim=1)
                stats.append(torch.stack([n.new_zeros(()), n.max() / n.median().clamp_min(1e-12)]))
        gain, row = (float(v) for v in torch.stack(stats).amax(dim=0).tolist())
        choice = "fp16x3" if (gain > self.AUTO_GAIN_RATIO or row > self.AUTO_ROW_RATIO) else DEFAULT_PRECISION
        self.precision_report = {"layernorm_gain_ratio": gain, "row_norm_ratio": row, "precision": choice}
        return choice

    # the measured part of the automatic mode: f16f8 is kept only if, on a probe batch, it stays this close (max-abs) to the split-fp16 mode,
    # which itself sits at the reference's own fp32 noise level.  The north-star bound is 1e-3; a quarter of it leaves room for inputs that
    # excite the weights more than the probe does.
    PROBE_TOL = 2.5e-4

    def _probe_features(self) -> torch.Tensor:
        """Two deterministic probe clips as input features [2, n_mels, T]: a piano-note clip of the synthetic set and broadband noise."""
        from . import synth
        from .feature_extraction import logmel_whisper_device
        from .weights import unit_variates
        n = self.cfg.n_frames * 160
        extra = self.probe_clips[:6]                                  # the probe stays a handful of clips
        pcm = np.zeros((2 + len(extra), n), dtype=np.float32)
        clip = synth.pcm_i16_to_f32(synth.synth_clips_i16(1, seed=1234, first=0)[0])
        m = min(n, clip.size)
        pcm[0, :m] = clip[:m]
        pcm[1, :m] = (0.1 * unit_variates("precision_probe", m, 0)).astype(np.float32)
        for i, c in enumerate(extra):                                 # the caller's audio, zero-padded / truncated like any clip
            k = min(n, c.size)
            pcm[2 + i, :k] = c[:k]
            m = max(m, k)
        return logmel_whisper_device(torch.from_numpy(pcm).to(self.device), max_valid=m, n_frames=self.cfg.n_frames, n_mels=self.cfg.n_mels)

    def _decide_precision(self) -> str:
        """precision=None: the weight statistics (choose_precision) can only ESCALATE to split-fp16; when they see nothing, the answer is
        measured: one probe batch through f16f8 and through fp16x3 on these very weights, f16f8 kept if the two agree to PROBE_TOL.  The
        winner's library handle (weights already uploaded) is kept; cost: one extra weight upload and two 2-clip forwards per load."""
        choice = self.choose_precision()
        if choice != DEFAULT_PRECISION:
            self.precision_report["decided_by"] = "weight statistics"
            return choice
        feats = self._probe_features()
        outs, handles = {}, {}
        L = _lib.lib()
        for prec in ("fp16x3", DEFAULT_PRECISION):
            self.precision = prec
            self._handle = None
            self._synced.clear()
            self._create_handle()
            self._push_all()
            ws = _lib.workspace(L.awt_encoder_workspace_bytes(self._handle, feats.shape[0]), self.device)
            out = torch.empty((feats.shape[0], self.cfg.max_source_positions, self.cfg.d_model), dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(L.awt_encoder_forward(self._handle, _lib.ptr(feats), feats.shape[0], feats.shape[2], _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                                 _lib.stream_handle()))
            outs[prec], handles[prec] = out, (self._handle, dict(self._synced))
        dist = float((outs[DEFAULT_PRECISION] - outs["fp16x3"]).abs().max())
        finite = bool(torch.isfinite(outs[DEFAULT_PRECISION]).all())
        choice = DEFAULT_PRECISION if (finite and dist < self.PROBE_TOL) else "fp16x3"
        per_clip = (outs[DEFAULT_PRECISION] - outs["fp16x3"]).abs().flatten(1).amax(dim=1).tolist()
        self.precision_report.update({"probe_max_abs_f16f8_vs_fp16x3": dist, "probe_tolerance": self.PROBE_TOL, "precision": choice, "decided_by": "probe",
                                      "probe_clips": {"synthetic": 2, "caller": int(feats.shape[0]) - 2}, "probe_max_abs_per_clip": [float(x) for x in per_clip]})
        for prec, (h, synced) in handles.items():
            if prec == choice:
                self._handle, self._synced = h, synced
            else:
                L.awt_encoder_destroy(h)
        return choice

    def _drop_handle(self):
        if self._handle is not None:
            _lib.lib().awt_encoder_destroy(self._handle)
            self._handle = None
        self._synced.clear()
        self._ws = None

    def _ensure_handle(self):
        if self._handle is not None:
            return
        if self._auto_precision:
            self.precision = self._decide_precision()        # the probe leaves the chosen mode's handle in place, weights uploaded
        if self._handle is None:
            self._create_handle()

    def _push_all(self) -> int:
        """Upload every parameter to the current handle (records the versions pushed)."""
        L = _lib.lib()
        n = 0
        with torch.cuda.device(self.device):
            for name, p in self.named_parameters():
                t = p.detach()
                if t.dtype != torch.float32 or not t.is_contiguous():
                    t = t.float().contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                _lib.check(L.awt_encoder_set_weight(self._handle, name.encode(), _lib.ptr(t), shape, t.dim(), _lib.stream_handle()))
                self._synced[name] = p._version
                n += 1
        return n

    def _create_handle(self):
        L = _lib.lib()
        bits = 0
        if self.lora is not None:
            for t in self.lora.targets:
                bits |= _lib.LORA_BITS[t]
        cfg = _lib.EncoderCfg(self.cfg.d_model, self.cfg.layers, self.cfg.heads, self.cfg.ffn, self.cfg.n_mels,
                              self.cfg.max_source_positions, PRECISIONS[self.precision],
                              self.lora.r if self.lora else 0, float(self.lora.alpha) if self.lora else 0.0, bits, self._chunk,
                              1 if self.trainable else 0, PRECISIONS[self.backward_precision] if self.backward_precision else 0)
        out = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.awt_encoder_create(_lib.ctx(self.device), C.byref(cfg), C.byref(out)))
        self._handle = out.value

    def sync_weights(self, force: bool = False) -> int:
        """Push every parameter that changed since the last push (tracked by tensor version) to the library."""
        fresh = self._handle is None       # the mode has just been decided from the current weights
        self._ensure_handle()
        L = _lib.lib()
        n = 0
        # walking the module tree costs more than the version checks, so the (name, parameter) list is cached; every leaf bumps a shared
        # epoch when one of its Parameter OBJECTS is replaced (`layer.fc1.weight = nn.Parameter(...)`, pruning / parametrize utilities),
        # which neither load_state_dict nor _apply sees -- a stale list would keep checking the OLD tensors' versions and run on stale weights
        if self._param_cache is None or self._param_ids != self._epoch[0]:
            self._param_cache = list(self.named_parameters())
            want = {n for n, _ in encoder_param_shapes(self.cfg)} | ({n for n, _ in lora_param_shapes(self.cfg, self.lora)} if self.lora is not None else set())
            have = {n for n, _ in self._param_cache}
            if have != want:                                   # a parameter was deleted / added behind the module's back: the library would keep stale planes
                raise KeyError(f"encoder parameters changed: missing {sorted(want - have)[:4]}, unexpected {sorted(have - want)[:4]}")
            if self._param_ids is not None and self._param_ids != self._epoch[0]:
                self._synced.clear()
            self._param_ids = self._epoch[0]
        if not force and all(self._synced.get(name) == p._version for name, p in self._param_cache):
            return 0
        if self._auto_precision and not fresh and any(self._synced.get(name) != p._version and "lora_" not in name for name, p in self._param_cache):
            self._drop_handle()                                # new base weights (load_state_dict after a forward): decide again, on them
            self._ensure_handle()
        with torch.cuda.device(self.device):
            for name, p in self._param_cache:
                ver = p._version
                if not force and self._synced.get(name) == ver:
                    continue
                t = p.detach()
                if t.dtype != torch.float32 or not t.is_contiguous():
                    t = t.float().contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                _lib.check(L.awt_encoder_set_weight(self._handle, name.encode(), _lib.ptr(t), shape, t.dim(), _lib.stream_handle()))
                self._synced[name] = ver
                n += 1
        return n

    def exact16_matrices(self):
        """(projection matrices whose weights are all fp16-representable, projection matrices): what the library found at upload time and
        therefore which GEMM form each matrix runs (include/awt.h: awt_encoder_exact16_matrices)."""
        self.sync_weights()
        ex, tot = C.c_int(), C.c_int()
        _lib.check(_lib.lib().awt_encoder_exact16_matrices(self._handle, C.byref(ex), C.byref(tot)))
        return ex.value, tot.value

    # ------------------------------------------------------------------------------------------------ gradients
    def lora_parameters_library_order(self) -> list:
        """Adapter parameters in the order awt_encoder_backward lays their gradients out: per layer, per target in
        (q_proj, k_proj, v_proj, out_proj, fc1, fc2) order: lora_A then lora_B (include/awt.h)."""
        order = [t for t in ("q_proj", "k_proj", "v_proj", "out_proj", "fc1", "fc2") if self.lora is not None and t in self.lora.targets]
        params = []
        for i in range(self.cfg.layers):
            layer = getattr(self.layers, str(i))
            for t in order:
                leaf = getattr(layer, t) if t in ("fc1", "fc2") else getattr(getattr(layer, "self_attn"), t)
                params += [leaf.lora_A, leaf.lora_B]
        return params

    def lora_grad_count(self) -> int:
        self._ensure_handle()
        return int(_lib.lib().awt_encoder_lora_grad_count(self._handle))

    def bind_grad_buffer(self, flat: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Bind the adapter gradients to ONE flat fp32 device buffer (library order): every adapter parameter's `.grad`
        becomes a view of it, the native backward writes there directly (accumulating from the second backward on, until
        `zero_adapter_grads()`), and the data-parallel exchange reduces it in place (dist.FlatGradBucket / set_comm)."""
        if not self.trainable:
            raise ValueError("bind_grad_buffer needs trainable=True")
        n = self.lora_grad_count()
        if flat is None:
            flat = torch.zeros(n, dtype=torch.float32, device=self.device)
        if flat.numel() != n or flat.dtype != torch.float32 or not flat.is_contiguous() or flat.device != self.device:
            raise ValueError(f"flat must be a contiguous float32 tensor of {n} elements on {self.device}")
        self._grad_flat = flat
        self._grad_params = self.lora_parameters_library_order()
        self._grad_views, off = [], 0
        for p in self._grad_params:
            k = p.numel()
            self._grad_views.append(flat[off: off + k].view_as(p))
            p.grad = self._grad_views[-1]
            off += k
        self._grad_fresh = True
        return flat

    def zero_adapter_grads(self) -> None:
        """With a bound buffer: the next backward overwrites instead of accumulating (no kernel is launched)."""
        self._grad_fresh = True

    def set_comm(self, comm, groups: int = 2) -> None:
        """Average the adapter gradients over `comm`'s ranks (dist.AwtComm) inside the backward call, in `groups` layer
        groups on the communicator's side stream (include/awt.h: awt_encoder_set_comm, AWT_BWD_ALLREDUCE)."""
        self._ensure_handle()
        _lib.check(_lib.lib().awt_encoder_set_comm(self._handle, None if comm is None else comm.handle, groups))
        self._comm = comm

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        res = super().load_state_dict(state_dict, strict=strict, **kw)
        self._synced.clear()
        self._param_cache = None
        return res

    def _apply(self, fn, *a, **kw):          # .to() / .cuda() / .float() replace parameter tensors
        self._param_cache = None
        self._synced.clear()
        return super()._apply(fn, *a, **kw)

    def close(self) -> None:
        """Release the library handle now (also done when the module is garbage-collected)."""
        if self._handle is not None:
            _lib.lib().awt_encoder_destroy(self._handle)
            self._handle = None
            self._synced.clear()

    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != self.device:
            self._ws = _lib.workspace(nbytes, self.device)
        return self._ws

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().awt_encoder_destroy(self._handle)
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, input_features: torch.Tensor, attention_mask=None, **kwargs) -> BaseModelOutput:
        if input_features.dim() != 3 or input_features.shape[1] != self.cfg.n_mels:
            raise ValueError(f"input_features must be [B, {self.cfg.n_mels}, T]")
        x = input_features.to(device=self.device, dtype=torch.float32).contiguous()
        B, _, T = x.shape
        self.sync_weights()
        L = _lib.lib()
        if self.trainable and torch.is_grad_enabled():
            params = self.lora_parameters_library_order()
            return BaseModelOutput(last_hidden_state=_EncoderLoRAFunction.apply(self, x, *params))
        ws = self._workspace(L.awt_encoder_workspace_bytes(self._handle, B))
        out = torch.empty((B, self.cfg.max_source_positions, self.cfg.d_model), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(L.awt_encoder_forward(self._handle, _lib.ptr(x), B, T, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                             _lib.stream_handle()))
        return BaseModelOutput(last_hidden_state=out)

    def encode_pcm(self, pcm: torch.Tensor, n_valid: Optional[torch.Tensor] = None, max_valid: Optional[int] = None,
                   return_features: bool = False):
        """Device PCM [B, n] (int16 / float32) -> last_hidden_state [B, S, d] (and input_features if asked): log-mel and
        encoder in one library call (`awt_audio_encode`), no host round trip."""
        if pcm.dim() != 2 or pcm.dtype not in (torch.int16, torch.float32) or not pcm.is_cuda:
            raise ValueError("pcm must be a [B, n] int16 or float32 device tensor")
        pcm = pcm.contiguous()
        B, n = pcm.shape
        max_valid = n if max_valid is None else int(max_valid)
        if n_valid is not None:
            n_valid = n_valid.to(device=pcm.device, dtype=torch.int32).contiguous()
        self.sync_weights()
        L = _lib.lib()
        ws = self._workspace(L.awt_audio_encode_workspace_bytes(self._handle, B))
        S, d, T = self.cfg.max_source_positions, self.cfg.d_model, self.cfg.n_frames
        out = torch.empty((B, S, d), dtype=torch.float32, device=pcm.device)
        feats = torch.empty((B, self.cfg.n_mels, T), dtype=torch.float32, device=pcm.device) if return_features else None
        with torch.cuda.device(pcm.device):
            _lib.check(L.awt_audio_encode(self._handle, _lib.ptr(pcm), int(pcm.dtype == torch.int16), pcm.stride(0),
                                          _lib.ptr(n_valid), max_valid, B, _lib.ptr(feats), _lib.ptr(out), _lib.ptr(ws),
                                          ws.numel(), _lib.stream_handle()))
        return (out, feats) if return_features else out
