"""ctypes binding of libawt.so (include/awt.h).  There is no fallback: if the library is missing or a call
fails, an exception is raised -- the product path never computes on the CPU."""
from __future__ import annotations

import ctypes as C
import os
import re
import threading
from typing import Optional

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AWT_LIB") or os.path.join(HERE, "libawt.so")   # AWT_LIB: kernel-variant A/B builds (tools/)
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "awt.h")

AWT_OK = 0
AWT_ERR_VALUE = -5


class AwtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libawt error {code}: {msg}")
        self.code = code
        self.msg = msg


class EncoderCfg(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32), ("ffn_dim", C.c_int32),
                ("n_mels", C.c_int32), ("n_ctx", C.c_int32), ("mfma_terms", C.c_int32), ("lora_rank", C.c_int32),
                ("lora_alpha", C.c_float), ("lora_targets", C.c_uint32), ("chunk_clips", C.c_int32), ("training", C.c_int32), ("backward_terms", C.c_int32)]


MFMA_PER_PAIR = {"bf16": 1, "fp16": 1, "bf16x3": 3, "fp16x3": 3, "f16f8": 2}   # MFMA-equivalents issued per fragment pair, by precision mode
BWD_ACCUMULATE, BWD_ALLREDUCE = 1, 2
COMM_ID_BYTES = 128
LORA_BITS = {"q_proj": 1, "k_proj": 2, "v_proj": 4, "out_proj": 8, "fc1": 16, "fc2": 32}
PROF_CLASSES = {"logmel": 0, "gemm": 1, "attention": 2, "layernorm": 3, "other": 4, "attention_bwd": 5}

_vp, _i, _i64, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float
_SIGNATURES = {
    "awt_last_error": (C.c_char_p, []),
    "awt_version": (C.c_char_p, []),
    "awt_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "awt_ctx_destroy": (None, [_vp]),
    "awt_logmel_workspace_bytes": (_sz, [_i]),
    "awt_logmel_whisper": (_i, [_vp, _vp, _i, _i64, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "awt_logmel_whisper_mels": (_i, [_vp, _vp, _i, _i64, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "awt_logmel_generic": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _vp]),
    "awt_logmel_prepare": (_i, [_vp, _i, _i, _f, _f, _i, _i]),
    "awt_resample_prepare": (_i, [_vp, _i, _i]),
    "awt_resampled_length": (_i64, [_i, _i, _i]),
    "awt_prepare_waveform": (_i, [_vp, _vp, _i, _i, _i64, _i64, _i, _i, _i, _vp, _i, _vp]),
    "awt_encoder_create": (_i, [_vp, C.POINTER(EncoderCfg), C.POINTER(_vp)]),
    "awt_encoder_destroy": (None, [_vp]),
    "awt_encoder_set_weight": (_i, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i, _vp]),
    "awt_encoder_exact16_matrices": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "awt_encoder_workspace_bytes": (_sz, [_vp, _i]),
    "awt_encoder_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "awt_encoder_train_workspace_bytes": (_sz, [_vp, _i]),
    "awt_encoder_lora_grad_count": (_sz, [_vp]),
    "awt_encoder_forward_train": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "awt_encoder_backward": (_i, [_vp, _vp, _i, _vp, _sz, _vp, _sz, _vp]),
    "awt_encoder_backward_ex": (_i, [_vp, _vp, _i, _vp, _sz, _vp, _sz, C.c_uint32, _vp]),
    "awt_comm_unique_id": (_i, [_vp]),
    "awt_comm_create": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "awt_comm_destroy": (None, [_vp]),
    "awt_comm_world": (_i, [_vp]),
    "awt_allreduce_sum_f32": (_i, [_vp, _vp, _sz, _vp]),
    "awt_allreduce_mean_f32": (_i, [_vp, _vp, _sz, _vp]),
    "awt_comm_bucket_stats": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_int64), _i, C.POINTER(_i)]),
    "awt_encoder_set_comm": (_i, [_vp, _vp, _i]),
    "awt_audio_encode_workspace_bytes": (_sz, [_vp, _i]),
    "awt_audio_encode": (_i, [_vp, _vp, _i, _i64, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "awt_op_linear": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "awt_op_linear_workspace_bytes": (_sz, [_i, _i, _i]),
    "awt_op_layernorm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "awt_op_attention": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "awt_op_attention_workspace_bytes": (_sz, [_i, _i, _i]),
    "awt_weight_create": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, C.POINTER(_vp)]),
    "awt_weight_destroy": (None, [_vp]),
    "awt_weight_padded_rows": (_i, [_vp]),
    "awt_linear_workspace_bytes": (_sz, [_vp, _i, _i]),
    "awt_linear_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "awt_linear_backward_input": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "awt_encoder_set_grad_scale_log2": (_i, [_vp, _i]),
    "awt_bmm_packed_bytes": (_sz, [_i, _i, _i]),
    "awt_bmm_pack": (_i, [_vp, _vp, _i64, _i64, _i, _i, _i, _vp, _sz, _vp]),
    "awt_bmm_workspace_bytes": (_sz, [_i, _i, _i]),
    "awt_bmm": (_i, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i, _i, _i, _i, _vp, _sz, _vp]),
    "awt_bmm_pack_kmajor": (_i, [_vp, _vp, _vp, _i, _i64, _i64, _i, _i, _i, _vp, _sz, _vp]),
    "awt_bmm_kmajor": (_i, [_vp, _vp, _vp, _i, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i, _i, _i, _i, _vp, _sz, _vp]),
    "awt_op_softmax_rows": (_i, [_vp, _vp, _vp, _i, _i, _i64, _f, _vp]),
    "awt_op_softmax_rows_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, _f, _vp]),
    "awt_op_embed": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "awt_op_gelu": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "awt_op_gelu_backward": (_i, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "awt_op_layernorm_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "awt_op_param_grad_workspace_bytes": (_sz, [_i, _i]),
    "awt_op_layernorm_param_grad": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _sz, _vp]),
    "awt_op_column_sums": (_i, [_vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "awt_op_cross_entropy": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "awt_op_attention_small": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "awt_op_attention_small_backward": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "awt_op_attention_small_dropout": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, C.c_float, C.c_uint64, _vp]),
    "awt_op_attention_small_backward_dropout": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, C.c_float, C.c_uint64, _vp]),
    "awt_tuning_set": (_i, [C.c_char_p, _i]),
    "awt_prof_enable": (_i, [_vp, _i]),
    "awt_prof_collect": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double)]),
}

_lib = None
_lock = threading.Lock()
_ctx = {}


def declared_symbols() -> list:
    """Function names declared in include/awt.h (used by the CPU-side export test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(awt_[a-z0-9_]+)\s*\(", text)))


def lib() -> C.CDLL:
    """Loads libawt.so (no GPU needed to load it); raises if it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(f"{LIB_PATH} is missing: run `python -m mlx8_ws_audio_transformer_amd.build` "
                                  "(there is no CPU fallback for this path)")
            l = C.CDLL(LIB_PATH)
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(l, name)
                fn.restype = res
                fn.argtypes = args
            _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != AWT_OK:
        msg = lib().awt_last_error().decode("utf-8", "replace")
        if rc == AWT_ERR_VALUE:
            raise ValueError(msg)  # the reference raises ValueError at the same place
        raise AwtError(rc, msg)


def ctx(device: Optional[torch.device] = None) -> int:
    """One awt_ctx per (process, device), created lazily; requires a visible gfx950 GPU."""
    if not torch.cuda.is_available():
        raise RuntimeError("mlx8_ws_audio_transformer_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                           "and this package has no CPU fallback")
    idx = torch.cuda.current_device() if device is None or device.index is None else device.index
    with _lock:
        h = _ctx.get(idx)
    if h is None:
        out = _vp()
        check(lib().awt_ctx_create(idx, C.byref(out)))
        with _lock:
            _ctx[idx] = out.value
        h = out.value
    return h


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("libawt needs contiguous device tensors")
    return t.data_ptr()


def stream_handle() -> int:
    return torch.cuda.current_stream().cuda_stream


def workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def prof_enable(on, classes=None) -> None:
    """Event-time the given kernel classes (names from PROF_CLASSES; default all) or switch timing off."""
    mask = 0
    if on:
        for k in (classes or PROF_CLASSES):
            mask |= 1 << PROF_CLASSES[k]
    check(lib().awt_prof_enable(ctx(), mask))


def prof_collect(klass: str):
    ms, n, fl = C.c_double(), _i64(), C.c_double()
    check(lib().awt_prof_collect(ctx(), PROF_CLASSES[klass], C.byref(ms), C.byref(n), C.byref(fl)))
    return ms.value, n.value, fl.value


def tuning_set(key: str, value: int) -> None:
    """Process-wide tuning / test hook (include/awt.h: awt_tuning_set), e.g. tuning_set("gemm_tile", 128)."""
    check(lib().awt_tuning_set(key.encode(), int(value)))
