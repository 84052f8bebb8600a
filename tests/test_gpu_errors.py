"""GPU: error behaviour of the C-ABI (codes + messages, nothing thrown across the boundary) and degenerate inputs."""
import ctypes as C

import numpy as np
import pytest
import torch

from mlx8_ws_audio_transformer_amd import _lib, weights as wts

pytestmark = pytest.mark.gpu


def _create(cfg, terms=3, **kw):
    L = _lib.lib()
    c = _lib.EncoderCfg(cfg.d_model, cfg.layers, cfg.heads, cfg.ffn, cfg.n_mels, cfg.max_source_positions, terms,
                        kw.get("r", 0), kw.get("alpha", 0.0), kw.get("targets", 0), 0, kw.get("training", 0))
    out = C.c_void_p()
    rc = L.awt_encoder_create(_lib.ctx(), C.byref(c), C.byref(out))
    return rc, out


def test_create_rejects_unsupported_configs():
    L = _lib.lib()
    rc, _ = _create(wts.EncoderConfig(64, 2, 2, 256, 80, 1500, "micro"))       # d_model 64
    assert rc == -1 and b"d_model" in L.awt_last_error()
    rc, _ = _create(wts.EncoderConfig(128, 2, 4, 512, 80, 1500, "hd32"))        # head_dim 32
    assert rc == -1 and b"head_dim" in L.awt_last_error()
    rc, _ = _create(wts.config("mini"), terms=7)
    assert rc == -1 and b"mfma_terms" in L.awt_last_error()
    rc, _ = _create(wts.config("mini"), terms=2, r=8, alpha=16.0, targets=_lib.LORA_BITS["fc1"], training=1)   # the single-fp16 measurement mode is inference-only too
    assert rc == -1 and b"mfma_terms must be 1 or 3" in L.awt_last_error()
    rc, _ = _create(wts.config("mini"), terms=5, r=8, alpha=16.0, targets=_lib.LORA_BITS["fc1"], training=1)   # f16f8 is inference-only
    assert rc == -1 and b"mfma_terms must be 1 or 3" in L.awt_last_error()
    rc, _ = _create(wts.config("mini"), training=1)                                                              # nothing to train
    assert rc == -1 and b"needs adapters" in L.awt_last_error()
    rc, h = _create(wts.config("mini"), r=8, alpha=16.0, targets=_lib.LORA_BITS["fc1"], training=1)              # every target trains
    assert rc == 0
    L.awt_encoder_destroy(h)


def test_set_weight_and_forward_state_errors():
    L = _lib.lib()
    cfg = wts.config("mini", True)
    rc, h = _create(cfg)
    assert rc == 0
    x = torch.zeros(128, 128, device="cuda")
    shape = (C.c_int64 * 2)(128, 128)
    assert L.awt_encoder_set_weight(h, b"layers.0.self_attn.nope.weight", _lib.ptr(x), shape, 2, None) == -1
    assert b"unknown parameter" in L.awt_last_error()
    assert L.awt_encoder_set_weight(h, b"layers.9.fc1.weight", _lib.ptr(x), shape, 2, None) == -1
    bshape = (C.c_int64 * 1)(128)
    assert L.awt_encoder_set_weight(h, b"layers.0.self_attn.k_proj.bias", _lib.ptr(x), bshape, 1, None) == -1
    assert b"k_proj has no bias" in L.awt_last_error()
    assert L.awt_encoder_set_weight(h, b"layers.0.fc1.weight", _lib.ptr(x), shape, 2, None) == -1     # [512, 128] expected
    assert b"expected [512,128]" in L.awt_last_error()
    assert L.awt_encoder_set_weight(h, b"layers.0.self_attn.q_proj.lora_A", _lib.ptr(x), shape, 2, None) == -4
    mel = torch.zeros(1, 80, 400, device="cuda")
    out = torch.empty(1, 200, 128, device="cuda")
    ws = _lib.workspace(L.awt_encoder_workspace_bytes(h, 1), "cuda")
    rc = L.awt_encoder_forward(h, _lib.ptr(mel), 1, 400, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None)
    assert rc == -4 and b"parameters uploaded" in L.awt_last_error()                                      # weights missing
    rc = L.awt_encoder_forward(h, _lib.ptr(mel), 1, 3000, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None)
    assert rc == -5 and b"length 400" in L.awt_last_error()                                               # the reference's ValueError
    L.awt_encoder_destroy(h)


def test_workspace_too_small_is_reported():
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    L = _lib.lib()
    enc = NativeWhisperEncoder(wts.config("mini", True)).eval()
    enc.sync_weights()
    mel = torch.zeros(2, 80, 400, device="cuda")
    out = torch.empty(2, 200, 128, device="cuda")
    ws = _lib.workspace(1024, "cuda")
    rc = L.awt_encoder_forward(enc._handle, _lib.ptr(mel), 2, 400, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None)
    assert rc == -3 and b"workspace too small" in L.awt_last_error()


def test_degenerate_clips():
    from mlx8_ws_audio_transformer_amd.feature_extraction import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor()
    out = fe([np.zeros(0, np.float32), np.ones(1, np.float32) * 0.5], sampling_rate=16000, return_tensors="np")["input_features"]
    assert out.shape == (2, 80, 3000) and np.all(out[0] == np.float32(-1.5)) and np.isfinite(out).all()
    from oracle import logmel
    np.testing.assert_allclose(out[1], logmel.whisper_logmel([np.ones(1, np.float32) * 0.5])[0], rtol=0, atol=1e-5)


def test_logmel_rejects_bad_arguments():
    L = _lib.lib()
    pcm = torch.zeros(2, 1000, dtype=torch.int16, device="cuda")
    out = torch.empty(2, 80, 3000, device="cuda")
    ws = _lib.workspace(L.awt_logmel_workspace_bytes(2), "cuda")
    rc = L.awt_logmel_whisper(_lib.ctx(), _lib.ptr(pcm), 1, 1000, None, 2000, 2, 3000, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None)
    assert rc == -1 and b"pcm_stride" in L.awt_last_error()
    rc = L.awt_logmel_whisper(_lib.ctx(), _lib.ptr(pcm), 1, 1000, None, 1000, 2, 3001, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None)
    assert rc == -1
    w = torch.zeros(1, 64000, device="cuda")
    o = torch.empty(1, 128, 126, device="cuda")
    rc = L.awt_logmel_generic(_lib.ctx(), _lib.ptr(w), 64000, 1, 64000, 16000, 1000, 512, 128, 0.0, 8000.0, 1e-6, _lib.ptr(o), None)
    assert rc == -1 and b"n_fft" in L.awt_last_error()
