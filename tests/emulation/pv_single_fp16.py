"""CPU estimate (fp64 + operand rounding) of the hidden-state error of ONE design change with everything else exact.  Not a test and not
product code: it imports oracle/ (allowed under tests/ only).  Run from the repository root: python tests/emulation/<this file> [tiny|small]."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import torch.nn.functional as F
from mlx8_ws_audio_transformer_amd import weights as wts, synth
from oracle import logmel as olm

torch.set_num_threads(8)
def f16(x): return x.to(torch.float16).to(x.dtype)
def f8round(x, bits=4):  # crude: keep `bits` significant bits (e4m3: 1+3)
    m, e = torch.frexp(x); return torch.ldexp(torch.round(m * 2**bits) / 2**bits, e)

def forward(W, mel, heads, mode, qgain=1.0):
    dt = torch.float64
    W = {k: torch.from_numpy(np.asarray(v)).to(dt) for k, v in W.items()}
    x = torch.from_numpy(np.asarray(mel)).to(dt)
    d = W["conv1.weight"].shape[0]; hd = d // heads
    h = F.gelu(F.conv1d(x, W["conv1.weight"], W["conv1.bias"], padding=1))
    h = F.gelu(F.conv1d(h, W["conv2.weight"], W["conv2.bias"], stride=2, padding=1))
    h = h.permute(0, 2, 1) + W["embed_positions.weight"]
    n_layers = 1 + max(int(k.split(".")[1]) for k in W if k.startswith("layers."))
    B, S_, _ = h.shape
    for i in range(n_layers):
        p = f"layers.{i}."
        y = F.layer_norm(h, (d,), W[p + "self_attn_layer_norm.weight"], W[p + "self_attn_layer_norm.bias"], 1e-5)
        q = (F.linear(y, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"]) * hd ** -0.5 * qgain).view(B, S_, heads, hd).transpose(1, 2)
        k = F.linear(y, W[p + "self_attn.k_proj.weight"]).view(B, S_, heads, hd).transpose(1, 2)
        v = F.linear(y, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"]).view(B, S_, heads, hd).transpose(1, 2)
        s = (f16(q) if mode == "q16" else q) @ k.transpose(2, 3)
        pu = torch.exp(s - s.amax(-1, keepdim=True))
        l = pu.sum(-1, keepdim=True)
        if mode in ("exact", "q16"): att = (pu @ v) / l
        elif mode == "p16v16": att = (f16(pu) @ f16(v)) / l
        elif mode == "p16v16_lr": att = (f16(pu) @ f16(v)) / f16(pu).sum(-1, keepdim=True)
        elif mode == "p16": att = (f16(pu) @ v) / l
        elif mode == "p16_vlo8":  # P16 V16 + P8 Vlo8
            vh = f16(v); att = (f16(pu) @ vh + f8round(pu) @ f8round(v - vh)) / l
        elif mode == "pbf16": att = (pu.to(torch.bfloat16).to(dt) @ v.to(torch.bfloat16).to(dt)) / l
        att = att.transpose(1, 2).reshape(B, S_, d)
        h = h + F.linear(att, W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"])
        y = F.layer_norm(h, (d,), W[p + "final_layer_norm.weight"], W[p + "final_layer_norm.bias"], 1e-5)
        y = F.gelu(F.linear(y, W[p + "fc1.weight"], W[p + "fc1.bias"]))
        h = h + F.linear(y, W[p + "fc2.weight"], W[p + "fc2.bias"])
    return F.layer_norm(h, (d,), W["layer_norm.weight"], W["layer_norm.bias"], 1e-5)

model = sys.argv[1] if len(sys.argv) > 1 else "tiny"
cfg = wts.config(model)
pcm = synth.synth_clips_i16(1, seed=1234)
mel = olm.whisper_logmel(pcm.astype(np.float32) / 32768.0) if hasattr(olm, "whisper_logmel") else None
print("mel", None if mel is None else mel.shape)
for profile, outl, qg in [("hf", False, 1.0), ("hf", False, 4.0), ("hf", False, 16.0), ("hf", True, 1.0), ("test", False, 1.0)]:
    W = wts.init_encoder_weights(cfg, 0, profile)
    if outl: W = wts.with_outlier_channels(W, cfg)
    t = time.time()
    ref = forward(W, mel, cfg.heads, "exact", qg)
    line = f"{model} {profile} outl={outl} qgain={qg}: |ref|max {float(ref.abs().max()):.1f}"
    for mode in ["p16v16", "q16", "p16", "pbf16"]:
        got = forward(W, mel, cfg.heads, mode, qg)
        line += f" | {mode} {float((got - ref).abs().max()):.2e}"
    print(line, f"({time.time() - t:.0f}s)", flush=True)
