"""Platform-independent deterministic parameter initialiser for Whisper-style encoders.

There is no network on the GPU box and no checkpoint on disk, so every benchmark and parity test
runs on random-init weights of the right architecture (SURVEY.md §8c "F4").  The values must be
bit-identical in the build container (where the HF oracle is available and the golden fixtures are
generated) and on the GPU box (where only this package travels), so the generator uses nothing but
64-bit integer hashing and exact IEEE multiplies -- no libm, no torch RNG streams:

    key   = fnv1a64(parameter name) ^ (seed * 0xD6E8FEB86659FD93)
    z_i   = splitmix64(key + i * 0x9E3779B97F4A7C15)                # counter based, i = flat index
    u     = sum of the four 16-bit fields of z_i                     # Irwin-Hall(4), integer
    value = (u - 131070) * (sqrt(3) / 65536) * std                   # zero mean, unit variance * std

Parameter names follow the HF state-dict keys of `WhisperEncoder`
(HF:models/whisper/modeling_whisper.py:555-577) so that any Whisper checkpoint that is present
locally loads into the native encoder unchanged, and so that the same dict loads into the HF
oracle with `load_state_dict` in tools/make_golden.py.
"""
from __future__ import annotations

import hashlib
import math
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_SEEDMIX = 0xD6E8FEB86659FD93
_MASK = (1 << 64) - 1


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def unit_variates(name: str, count: int, seed: int = 0) -> np.ndarray:
    """`count` zero-mean unit-variance float64 variates keyed by (seed, name, flat index)."""
    key = np.uint64((_fnv1a64(name) ^ ((seed * _SEEDMIX) & _MASK)) & _MASK)
    out = np.empty(count, dtype=np.float64)
    step = 1 << 22
    scale = math.sqrt(3.0) / 65536.0
    for lo in range(0, count, step):
        hi = min(count, lo + step)
        with np.errstate(over="ignore"):
            ctr = key + np.arange(lo, hi, dtype=np.uint64) * _GOLDEN
        z = _splitmix64(ctr)
        m = np.uint64(0xFFFF)
        u = (z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m) + (z >> np.uint64(48))
        out[lo:hi] = (u.astype(np.int64) - 131070).astype(np.float64) * scale
    return out


def decoder_param_shapes(d: int, layers: int, ffn: int, vocab: int, max_target_positions: int = 448) -> List[Tuple[str, Tuple[int, ...]]]:
    """HF `WhisperDecoder.state_dict()` keys and shapes (HF:modeling_whisper.py:416-507,649-700)."""
    shapes: List[Tuple[str, Tuple[int, ...]]] = [("embed_tokens.weight", (vocab, d)), ("embed_positions.weight", (max_target_positions, d))]
    for i in range(layers):
        p = f"layers.{i}."
        for att in ("self_attn", "encoder_attn"):
            shapes += [(p + att + ".k_proj.weight", (d, d)), (p + att + ".v_proj.weight", (d, d)), (p + att + ".v_proj.bias", (d,)),
                       (p + att + ".q_proj.weight", (d, d)), (p + att + ".q_proj.bias", (d,)),
                       (p + att + ".out_proj.weight", (d, d)), (p + att + ".out_proj.bias", (d,)),
                       (p + att + "_layer_norm.weight", (d,)), (p + att + "_layer_norm.bias", (d,))]
        shapes += [(p + "fc1.weight", (ffn, d)), (p + "fc1.bias", (ffn,)), (p + "fc2.weight", (d, ffn)), (p + "fc2.bias", (d,)),
                   (p + "final_layer_norm.weight", (d,)), (p + "final_layer_norm.bias", (d,))]
    return shapes + [("layer_norm.weight", (d,)), ("layer_norm.bias", (d,))]


def init_decoder_weights(d: int, layers: int, ffn: int, vocab: int, max_target_positions: int = 448, seed: int = 0) -> Dict[str, np.ndarray]:
    """Platform-independent fp32 decoder state dict (names prefixed "decoder." for the generator key): linears of std
    d^-1/2 / 2, token embeddings of std 0.1, positions of std 0.5, non-trivial biases and LayerNorm affines, so that logits are far from uniform."""
    out: Dict[str, np.ndarray] = {}
    for name, shape in decoder_param_shapes(d, layers, ffn, vocab, max_target_positions):
        n = int(np.prod(shape))
        u = unit_variates("decoder." + name, n, seed)
        if "layer_norm.weight" in name:
            v = 1.0 + 0.1 * u
        elif name.startswith("embed_"):
            v = (0.5 if "positions" in name else 0.1) * u     # strong positions: greedy decoding does not collapse to one token
        elif name.endswith(".weight"):
            v = u * (0.5 / math.sqrt(shape[1]))
        else:
            v = 0.05 * u
        out[name] = v.astype(np.float32).reshape(shape)
    return out


@dataclass(frozen=True)
class EncoderConfig:
    """Shape of a Whisper-style audio encoder (HF `WhisperConfig` field names in comments)."""

    d_model: int = 384                # d_model
    layers: int = 4                   # encoder_layers
    heads: int = 6                    # encoder_attention_heads
    ffn: int = 1536                   # encoder_ffn_dim
    n_mels: int = 80                  # num_mel_bins
    max_source_positions: int = 1500  # S; mel length T = 2*S
    name: str = "tiny"

    @property
    def head_dim(self) -> int:
        return self.d_model // self.heads

    @property
    def n_frames(self) -> int:
        return 2 * self.max_source_positions


CONFIGS: Dict[str, EncoderConfig] = {
    "micro": EncoderConfig(64, 2, 2, 256, 80, 1500, "micro"),      # heads of 32: tests non-64 head_dim rejection
    "mini": EncoderConfig(128, 2, 2, 512, 80, 1500, "mini"),       # head_dim 64, cheap enough for full-tensor goldens
    "tiny": EncoderConfig(384, 4, 6, 1536, 80, 1500, "tiny"),
    "base": EncoderConfig(512, 6, 8, 2048, 80, 1500, "base"),
    "small": EncoderConfig(768, 12, 12, 3072, 80, 1500, "small"),
    "medium": EncoderConfig(1024, 24, 16, 4096, 80, 1500, "medium"),
    "large": EncoderConfig(1280, 32, 20, 5120, 80, 1500, "large"),   # large / large-v2 (80 mel bins)
    "large-v3": EncoderConfig(1280, 32, 20, 5120, 128, 1500, "large-v3"),   # 128 mel bins (also the turbo checkpoints' encoder)
}


def config(name: str, trimmed: bool = False) -> EncoderConfig:
    """Named config; `trimmed=True` gives the T=400 / S=200 variant (not reference-equivalent)."""
    c = CONFIGS[name]
    if trimmed:
        c = EncoderConfig(c.d_model, c.layers, c.heads, c.ffn, c.n_mels, 200, c.name + "-trimmed")
    return c


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> np.ndarray:
    """Fixed positional table, `cat(sin, cos)` (HF:modeling_whisper.py:55-64). Computed in fp32 like HF."""
    import torch  # same op sequence as HF so the table is bit-identical to the oracle's

    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length).view(-1, 1) * inv.view(1, -1)
    return torch.cat([t.sin(), t.cos()], dim=1).numpy()


def encoder_param_shapes(cfg: EncoderConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    d, f = cfg.d_model, cfg.ffn
    shapes: List[Tuple[str, Tuple[int, ...]]] = [
        ("conv1.weight", (d, cfg.n_mels, 3)),
        ("conv1.bias", (d,)),
        ("conv2.weight", (d, d, 3)),
        ("conv2.bias", (d,)),
        ("embed_positions.weight", (cfg.max_source_positions, d)),
    ]
    for i in range(cfg.layers):
        p = f"layers.{i}."
        shapes += [
            (p + "self_attn.k_proj.weight", (d, d)),
            (p + "self_attn.v_proj.weight", (d, d)),
            (p + "self_attn.v_proj.bias", (d,)),
            (p + "self_attn.q_proj.weight", (d, d)),
            (p + "self_attn.q_proj.bias", (d,)),
            (p + "self_attn.out_proj.weight", (d, d)),
            (p + "self_attn.out_proj.bias", (d,)),
            (p + "self_attn_layer_norm.weight", (d,)),
            (p + "self_attn_layer_norm.bias", (d,)),
            (p + "fc1.weight", (f, d)),
            (p + "fc1.bias", (f,)),
            (p + "fc2.weight", (d, f)),
            (p + "fc2.bias", (d,)),
            (p + "final_layer_norm.weight", (d,)),
            (p + "final_layer_norm.bias", (d,)),
        ]
    shapes += [("layer_norm.weight", (d,)), ("layer_norm.bias", (d,))]
    return shapes


def init_encoder_weights(cfg: EncoderConfig, seed: int = 0, profile: str = "hf") -> Dict[str, np.ndarray]:
    """fp32 state dict for `cfg`.

    profile "hf":   HF `_init_weights` statistics -- N(0, 0.02^2)-like linears/convs, zero biases,
                    LayerNorm gamma 1 / beta 0 (what BASELINE.md §4 / SURVEY.md §8d ask the bench to use).
    profile "test": same weights but every bias and LayerNorm affine is non-trivial and q/k projections
                    are scaled to sqrt(2/d), so that bias paths, the q pre-scale and a non-uniform softmax are
                    actually exercised by the parity tests.
    """
    if profile not in ("hf", "test"):
        raise ValueError(f"unknown init profile {profile!r}")
    out: Dict[str, np.ndarray] = {}
    for name, shape in encoder_param_shapes(cfg):
        n = int(np.prod(shape))
        if name == "embed_positions.weight":
            out[name] = sinusoids(*shape).astype(np.float32)
            continue
        is_ln = "layer_norm" in name
        if name.endswith(".weight") and not is_ln:
            std = 0.02
            if profile == "test" and (".q_proj." in name or ".k_proj." in name):
                std = math.sqrt(2.0 / cfg.d_model)   # attention logits of std ~2 instead of ~0.03
            v = unit_variates(name, n, seed) * std
        elif is_ln and name.endswith(".weight"):
            v = np.ones(n) if profile == "hf" else 1.0 + 0.1 * unit_variates(name, n, seed)
        else:  # biases (linear, conv, LayerNorm beta)
            v = np.zeros(n) if profile == "hf" else 0.05 * unit_variates(name, n, seed)
        out[name] = v.astype(np.float32).reshape(shape)
    return out


@dataclass(frozen=True)
class LoraSpec:
    """Build-defined LoRA adapters: y = xW^T + b + (alpha/r) (x A^T) B^T."""

    r: int = 8
    alpha: float = 16.0
    targets: Tuple[str, ...] = ("q_proj", "v_proj")

    @property
    def scale(self) -> float:
        return self.alpha / self.r


def lora_param_shapes(cfg: EncoderConfig, spec: LoraSpec) -> List[Tuple[str, Tuple[int, ...]]]:
    d, f = cfg.d_model, cfg.ffn
    dims = {"q_proj": (d, d), "k_proj": (d, d), "v_proj": (d, d), "out_proj": (d, d), "fc1": (f, d), "fc2": (d, f)}
    shapes = []
    for i in range(cfg.layers):
        for t in spec.targets:
            mod = f"layers.{i}." + (f"self_attn.{t}" if t.endswith("_proj") else t)
            dout, din = dims[t]
            shapes.append((mod + ".lora_A", (spec.r, din)))
            shapes.append((mod + ".lora_B", (dout, spec.r)))
    return shapes


def init_lora_weights(cfg: EncoderConfig, spec: LoraSpec, seed: int = 0, zero_b: bool = True) -> Dict[str, np.ndarray]:
    """A ~ N(0, 1/r)-like, B = 0 at init (standard LoRA); `zero_b=False` gives a non-trivial B for parity tests."""
    out: Dict[str, np.ndarray] = {}
    for name, shape in lora_param_shapes(cfg, spec):
        n = int(np.prod(shape))
        if name.endswith("lora_A"):
            v = unit_variates(name, n, seed) * (1.0 / math.sqrt(shape[1]))
        else:
            v = np.zeros(n) if zero_b else unit_variates(name, n, seed) * 0.02
        out[name] = v.astype(np.float32).reshape(shape)
    return out


def with_mlp_outliers(W: Dict[str, np.ndarray], cfg: "EncoderConfig", seed: int = 0, row_gain: float = 12.0, conv_gain: float = 8.0) -> Dict[str, np.ndarray]:
    """Adversarial profile for the automatic precision choice: outliers where `choose_precision` does NOT look -- a few fc1 output rows (huge
    MLP hidden units feeding fc2) and a few conv2 output channels (large residual-stream channels from the stem on).  LayerNorm gains and
    out_proj / fc2 row norms stay ordinary, so only a measurement can tell what f16f8 makes of these weights."""
    W = dict(W)
    rng = np.random.default_rng(seed + 17)
    for name in list(W):
        if name.endswith("fc1.weight"):
            idx = rng.choice(W[name].shape[0], 4, replace=False)
            W[name] = W[name].copy(); W[name][idx] *= row_gain
            b = name[:-len("weight")] + "bias"
            W[b] = W[b].copy(); W[b][idx] *= row_gain
        if name == "conv2.weight":
            idx = rng.choice(W[name].shape[0], 4, replace=False)
            W[name] = W[name].copy(); W[name][idx] *= conv_gain
    return W


def with_peaked_attention(W: Dict[str, np.ndarray], cfg: "EncoderConfig", qk_gain: float = 6.0, v_gain: float = 12.0) -> Dict[str, np.ndarray]:
    """Adversarial profile for the single-product P V of the f16f8 attention: sharp attention (q / k projections scaled up, logits of tens)
    onto LARGE values (v projection and its bias scaled up): the P V error is 2^-12 |v|_max per output, so this is where it shows.  Nothing
    `choose_precision` looks at changes."""
    W = dict(W)
    for name in list(W):
        if name.endswith("self_attn.q_proj.weight") or name.endswith("self_attn.k_proj.weight") or name.endswith("self_attn.q_proj.bias"):
            W[name] = W[name] * np.float32(qk_gain)
        if name.endswith("self_attn.v_proj.weight") or name.endswith("self_attn.v_proj.bias"):
            W[name] = W[name] * np.float32(v_gain)
    return W


def weights_digest(weights: Dict[str, np.ndarray]) -> str:
    """sha256 over the fp32 blobs in name order -- proves both sides built the same weights.

    The sinusoid table is skipped: it is the only entry that goes through libm (sin/cos/exp), so it
    may differ in the last bit between hosts; tests/golden pins it by value instead (fixture F5).
    """
    h = hashlib.sha256()
    for k in sorted(weights):
        if k == "embed_positions.weight":
            continue
        h.update(k.encode())
        h.update(np.ascontiguousarray(weights[k], dtype=np.float32).tobytes())
    return h.hexdigest()


def with_outlier_channels(W: Dict[str, np.ndarray], cfg: "EncoderConfig", seed: int = 0, ln_gain: float = 30.0, row_gain: float = 10.0) -> Dict[str, np.ndarray]:
    """"Realistic-outlier" weight profile: trained Whisper checkpoints carry a few very large LayerNorm gains and massive
    residual channels.  Four channels of every LayerNorm weight are multiplied by `ln_gain`, two output rows of every fc2 /
    out_proj weight by `row_gain`, everything else is left as is (no checkpoint exists offline: SURVEY.md section 8c).  Used by the
    parity tests and by bench.py's `outlier_profile` block to show what each operand precision keeps of the absolute 1e-3 bound."""
    W = dict(W)
    rng = np.random.default_rng(seed)
    for name in list(W):
        if name.endswith("_layer_norm.weight") or name == "layer_norm.weight":
            idx = rng.choice(cfg.d_model, 4, replace=False)
            W[name] = W[name].copy(); W[name][idx] *= ln_gain
        if name.endswith("fc2.weight") or name.endswith("out_proj.weight"):
            idx = rng.choice(W[name].shape[0], 2, replace=False)
            W[name] = W[name].copy(); W[name][idx] *= row_gain      # rows feeding outlier residual channels
    return W
