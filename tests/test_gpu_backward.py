"""GPU parity of the LoRA backward pass (native HIP, through the C-ABI) vs torch autograd on the fp32 oracle.
The LoRA term has no reference implementation: parity is pinned to the build's own restatement."""
import numpy as np
import pytest
import torch

from mlx8_ws_audio_transformer_amd import weights as wts
from oracle import encoder as oracle_enc
from oracle import logmel as oracle_mel
from tests.util import piano_clips_f32

pytestmark = pytest.mark.gpu


def _oracle_grads(W, LW, mel, cfg, spec, dout):
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    Lt = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in LW.items()}
    out = oracle_enc.encoder_forward({**Wt, **Lt}, mel, cfg.heads, lora_scale=spec.scale)
    (out * torch.from_numpy(dout)).sum().backward()
    return out.detach().numpy(), {k: v.grad.numpy() for k, v in Lt.items()}


ALL = ("q_proj", "k_proj", "v_proj", "out_proj", "fc1", "fc2")


@pytest.mark.parametrize("name,trimmed,targets,r", [("mini", True, ("q_proj", "v_proj"), 8), ("tiny", True, ("q_proj", "k_proj", "v_proj"), 16),
                                                    # adapters on every encoder linear, and groups without the q/k/v group
                                                    ("mini", True, ALL, 8), ("tiny", True, ALL, 16), ("mini", True, ("out_proj", "fc2"), 8), ("mini", False, ("v_proj", "fc1"), 4),
                                                    ("mini", False, ("q_proj", "v_proj"), 8),
                                                    # BASELINE.json configs[2] / [3] at full size: Whisper-small (d = 768, 12 layers), parity
                                                    # mode S = 1500, adapters on q_proj, v_proj, r = 8 and r = 16 (oracle autograd: ~20 s of CPU)
                                                    ("small", False, ("q_proj", "v_proj"), 8), ("small", False, ("q_proj", "v_proj"), 16)])
def test_lora_gradients_match_oracle_autograd(name, trimmed, targets, r):
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    cfg = wts.config(name, trimmed)
    spec = wts.LoraSpec(r=r, alpha=16.0, targets=targets)
    W = wts.init_encoder_weights(cfg, 0, "test")
    LW = wts.init_lora_weights(cfg, spec, 0, zero_b=False)
    B = 2
    mel = oracle_mel.whisper_logmel(piano_clips_f32(B), n_samples=cfg.n_frames * 160)
    dout = (wts.unit_variates("dout", B * cfg.max_source_positions * cfg.d_model, 3).reshape(B, cfg.max_source_positions, cfg.d_model)
            / np.sqrt(cfg.max_source_positions)).astype(np.float32)
    ref_out, ref_g = _oracle_grads(W, LW, mel, cfg, spec, dout)

    enc = NativeWhisperEncoder(cfg, precision="bf16x3", lora=spec, trainable=True, seed=0, init_profile="test")
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in {**W, **LW}.items()})
    out = enc(torch.from_numpy(mel).cuda()).last_hidden_state
    assert out.requires_grad
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out, rtol=0, atol=1e-3)
    (out * torch.from_numpy(dout).cuda()).sum().backward()
    worst = 0.0
    for k, g_ref in ref_g.items():
        mod, leaf = k.rsplit(".", 1)
        p = enc
        for part in mod.split("."):
            p = getattr(p, part)
        g = getattr(p, leaf).grad
        assert g is not None, k
        err = np.abs(g.cpu().numpy() - g_ref).max() / max(np.abs(g_ref).max(), 1e-12)
        worst = max(worst, err)
        assert err < 2e-3, (k, err)   # relative to the gradient's own scale
    assert all(p.grad is None for n, p in enc.named_parameters() if "lora_" not in n)
    print("worst relative gradient error", worst)


def test_backward_is_reproducible_and_rejects_untrainable_configurations():
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    cfg = wts.config("mini", True)
    with pytest.raises(ValueError):
        NativeWhisperEncoder(cfg, trainable=True)                                   # no adapters: nothing to train
    with pytest.raises(ValueError):
        NativeWhisperEncoder(cfg, lora=wts.LoraSpec(), trainable=True, precision="f16f8")   # training keeps bf16 planes
    spec = wts.LoraSpec(r=8, alpha=16.0, targets=ALL)
    enc = NativeWhisperEncoder(cfg, precision="bf16x3", lora=spec, trainable=True, seed=0, init_profile="test")
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in {**wts.init_encoder_weights(cfg, 0, "test"),
                                                              **wts.init_lora_weights(cfg, spec, 0, zero_b=False)}.items()})
    mel = torch.from_numpy(oracle_mel.whisper_logmel(piano_clips_f32(2), n_samples=cfg.n_frames * 160)).cuda()
    runs = []
    for _ in range(2):
        enc.zero_grad()
        enc(mel).last_hidden_state.square().mean().backward()
        runs.append(torch.cat([p.grad.flatten() for n, p in enc.named_parameters() if "lora_" in n]).clone())
    assert torch.equal(runs[0], runs[1])     # no atomics anywhere in the backward: bit-reproducible
    with torch.no_grad():
        assert not enc(mel).last_hidden_state.requires_grad


def test_backward_is_bit_identical_under_every_gemm_tiling():
    """Forward-for-training and backward (transposed frozen weights, GELU' epilogue, LoRA segments) under each forced block
    tiling: same adapter gradients bit for bit."""
    from mlx8_ws_audio_transformer_amd import _lib
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    cfg = wts.config("tiny", True)
    spec = wts.LoraSpec(r=8, alpha=16.0, targets=("q_proj", "v_proj"))
    LW = wts.init_lora_weights(cfg, spec, 0, zero_b=False)
    enc = NativeWhisperEncoder(cfg, precision="bf16x3", lora=spec, trainable=True, seed=0, init_profile="test")
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in LW.items()}, strict=False)
    mel = torch.from_numpy(oracle_mel.whisper_logmel(piano_clips_f32(3), n_samples=cfg.n_frames * 160)).cuda()
    grads = []
    for tile in (0, 64, 128, 256):
        _lib.tuning_set("gemm_tile", tile)
        try:
            for p in enc.parameters():
                p.grad = None
            out = enc(mel).last_hidden_state
            (out * out).sum().backward()
            grads.append(torch.cat([p.grad.flatten() for n, p in enc.named_parameters() if "lora_" in n]).clone())
        finally:
            _lib.tuning_set("gemm_tile", 0)
    assert float(grads[0].abs().max()) > 0
    for g in grads[1:]:
        assert torch.equal(g, grads[0])


def test_bf16_backward_option_stays_close_to_the_exact_backward():
    """backward_precision="bf16": dp / dq / dk / dv and the backward GEMMs use one bf16 product per fragment pair (scores are
    still recomputed in split-bf16).  The adapter gradients must stay within mixed-precision distance of the default."""
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    cfg = wts.config("tiny", True)
    spec = wts.LoraSpec(r=8, alpha=16.0, targets=("q_proj", "v_proj"))
    LW = wts.init_lora_weights(cfg, spec, 0, zero_b=False)
    mel = torch.from_numpy(oracle_mel.whisper_logmel(piano_clips_f32(2), n_samples=cfg.n_frames * 160)).cuda()
    grads, outs = {}, {}
    for bp in (None, "bf16"):
        enc = NativeWhisperEncoder(cfg, precision="bf16x3", lora=spec, trainable=True, seed=0, init_profile="test", backward_precision=bp)
        enc.load_state_dict({k: torch.from_numpy(v) for k, v in LW.items()}, strict=False)
        out = enc(mel).last_hidden_state
        (out * out).sum().backward()
        outs[bp] = out.detach().clone()
        grads[bp] = torch.cat([p.grad.flatten() for n, p in enc.named_parameters() if "lora_" in n])
    assert torch.equal(outs[None], outs["bf16"])                       # the forward is untouched
    rel = float((grads["bf16"] - grads[None]).norm() / grads[None].norm())
    assert 1e-5 < rel < 2e-2, rel
    with pytest.raises(ValueError):
        NativeWhisperEncoder(cfg, precision="bf16x3", lora=spec, trainable=True, backward_precision="fp8")


def test_lora_gradients_at_large_width():
    """d_model 1280 (Whisper large): the adapter-gradient reduction runs in 1024-column chunks."""
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    cfg = wts.EncoderConfig(1280, 2, 20, 5120, 80, 200, "large-2layer-trimmed")
    spec = wts.LoraSpec(r=8, alpha=16.0, targets=("q_proj", "v_proj"))
    W = wts.init_encoder_weights(cfg, 0, "test")
    LW = wts.init_lora_weights(cfg, spec, 0, zero_b=False)
    mel = oracle_mel.whisper_logmel(piano_clips_f32(1), n_samples=cfg.n_frames * 160)
    dout = (wts.unit_variates("dout", cfg.max_source_positions * cfg.d_model, 5).reshape(1, cfg.max_source_positions, cfg.d_model)
            / np.sqrt(cfg.max_source_positions)).astype(np.float32)
    _, ref_g = _oracle_grads(W, LW, mel, cfg, spec, dout)
    enc = NativeWhisperEncoder(cfg, precision="bf16x3", lora=spec, trainable=True, seed=0, init_profile="test")
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in {**W, **LW}.items()})
    out = enc(torch.from_numpy(mel).cuda()).last_hidden_state
    (out * torch.from_numpy(dout).cuda()).sum().backward()
    for name, p in enc.named_parameters():
        if "lora_" in name:
            ref = ref_g[name]
            err = float(np.abs(p.grad.cpu().numpy() - ref).max()) / max(float(np.abs(ref).max()), 1e-12)
            assert err < 1e-3, (name, err)


@pytest.mark.parametrize("name,trimmed,targets,r,gmag", [("mini", True, ("q_proj", "v_proj"), 8, 1.0), ("tiny", True, ("q_proj", "k_proj", "v_proj", "out_proj"), 16, 1e-6),
                                                         ("small", False, ("q_proj", "v_proj"), 8, 1e-3)])
def test_f16f8_mlp_backward_matches_oracle_autograd(name, trimmed, targets, r, gmag):
    """awt_encoder_cfg.backward_terms = 5 (`backward_precision="f16f8"`): the MLP's two backward GEMMs in the f16f8 operand format with a power-of-two
    gradient scale chosen from max |d loss / d hidden| (here tiny and large upstream gradients on purpose).  Same bound against the oracle's autograd as
    the split-bf16 backward, and close to it."""
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    cfg = wts.config(name, trimmed)
    spec = wts.LoraSpec(r=r, alpha=16.0, targets=targets)
    W = wts.init_encoder_weights(cfg, 0, "test")
    LW = wts.init_lora_weights(cfg, spec, 0, zero_b=False)
    B = 2
    mel = oracle_mel.whisper_logmel(piano_clips_f32(B), n_samples=cfg.n_frames * 160)
    dout = (gmag * wts.unit_variates("dout", B * cfg.max_source_positions * cfg.d_model, 3).reshape(B, cfg.max_source_positions, cfg.d_model)
            / np.sqrt(cfg.max_source_positions)).astype(np.float32)
    _, ref_g = _oracle_grads(W, LW, mel, cfg, spec, dout)
    got = {}
    for bp in (None, "f16f8"):
        enc = NativeWhisperEncoder(cfg, precision="bf16x3", lora=spec, trainable=True, seed=0, init_profile="test", backward_precision=bp)
        enc.load_state_dict({k: torch.from_numpy(v) for k, v in {**W, **LW}.items()})
        out = enc(torch.from_numpy(mel).cuda()).last_hidden_state
        (out * torch.from_numpy(dout).cuda()).sum().backward()
        got[bp] = {n: p.grad.cpu().numpy() for n, p in enc.named_parameters() if "lora_" in n}
    worst = worst_vs_bf16 = 0.0
    for k, g_ref in ref_g.items():
        scale = max(np.abs(g_ref).max(), 1e-30)
        worst = max(worst, np.abs(got["f16f8"][k] - g_ref).max() / scale)
        worst_vs_bf16 = max(worst_vs_bf16, np.abs(got["f16f8"][k] - got[None][k]).max() / scale)
    print("worst relative gradient error f16f8-MLP backward:", worst, " vs split-bf16 backward:", worst_vs_bf16)
    assert worst < 2e-3 and worst_vs_bf16 < 5e-4, (worst, worst_vs_bf16)
    assert worst_vs_bf16 > 0.0                              # the f16f8 GEMMs really ran


def test_f16f8_mlp_backward_rejects_unsupported_configurations():
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    cfg = wts.config("mini", True)
    with pytest.raises(ValueError):                          # adapters on the MLP are not part of this mode
        NativeWhisperEncoder(cfg, precision="bf16x3", lora=wts.LoraSpec(r=8, alpha=16.0, targets=("q_proj", "fc1")), trainable=True, backward_precision="f16f8")
    with pytest.raises(ValueError):                          # inference encoders have no backward
        NativeWhisperEncoder(cfg, precision="bf16x3", lora=wts.LoraSpec(r=8, alpha=16.0, targets=("q_proj",)), backward_precision="f16f8")


def test_f16f8_mlp_backward_saturates_instead_of_overflowing_on_an_outlier_gain():
    """the gradient scale 2^k of the f16f8 backward is chosen from max |d loss / d hidden| alone; an intermediate gradient far above it (LayerNorm
    backward through an outlier gain of a lower layer) used to overflow the fp16 gradient plane -> inf, -inf residual, NaN adapter gradients that AdamW
    would bake in.  The gradient planes now saturate at fp16's largest finite value: every adapter gradient stays finite, and the layers ABOVE the outlier
    (whose gradients never pass through its backward) stay close to the split-bf16 backward -- loosely: a 2500x gain also amplifies the 1e-5 relative
    difference between the two MLP operand formats of the layers below it into per-cent differences of everything downstream in the FORWARD pass."""
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    cfg = wts.config("tiny", True)
    spec = wts.LoraSpec(r=8, alpha=16.0, targets=("q_proj", "v_proj"))
    W = {k: v.copy() for k, v in wts.init_encoder_weights(cfg, 0, "test").items()}
    # an outlier gain on one channel of layer 2's attention LayerNorm: its BACKWARD multiplies the gradient that flows on into layer 1's MLP (the f16f8
    # backward GEMMs' input planes) by 2500 on that channel -- beyond the 2^10 of head-room the scale leaves above max |d loss / d hidden| -- while the forward
    # activations (|LN output| <= ~1.5e4) stay inside fp16's range
    W["layers.2.self_attn_layer_norm.weight"][7] = 2500.0
    LW = wts.init_lora_weights(cfg, spec, 0, zero_b=False)
    B = 2
    mel = oracle_mel.whisper_logmel(piano_clips_f32(B), n_samples=cfg.n_frames * 160)
    dout = (wts.unit_variates("dout", B * cfg.max_source_positions * cfg.d_model, 3).reshape(B, cfg.max_source_positions, cfg.d_model)
            / np.sqrt(cfg.max_source_positions)).astype(np.float32)
    got = {}
    for bp in (None, "f16f8"):
        enc = NativeWhisperEncoder(cfg, precision="bf16x3", lora=spec, trainable=True, seed=0, init_profile="test", backward_precision=bp)
        enc.load_state_dict({k: torch.from_numpy(v) for k, v in {**W, **LW}.items()})
        out = enc(torch.from_numpy(mel).cuda()).last_hidden_state
        (out * torch.from_numpy(dout).cuda()).sum().backward()
        got[bp] = {n: p.grad.cpu().numpy() for n, p in enc.named_parameters() if "lora_" in n}
    for n, g in got["f16f8"].items():
        assert np.isfinite(g).all(), n
    upper = [n for n in got[None] if any(n.startswith("layers.%d." % li) for li in (2, 3))]
    assert upper
    for n in upper:
        scale = max(np.abs(got[None][n]).max(), 1e-30)
        assert np.abs(got["f16f8"][n] - got[None][n]).max() / scale < 0.2, n
