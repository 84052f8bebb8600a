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
def qfp(x, ebits, mbits, bias, maxv):
    # generic small float quantiser with subnormals, RNE-ish (round half away is fine for an estimate)
    ax = x.abs().clamp(max=maxv)
    e = torch.floor(torch.log2(ax.clamp_min(1e-300)))
    emin = 1 - bias
    e = e.clamp(min=emin)
    step = torch.pow(2.0, e - mbits)
    return torch.sign(x) * torch.round(ax / step) * step
def e4m3(x): return qfp(x, 4, 3, 7, 448.0)
def e3m2(x): return qfp(x, 3, 2, 3, 28.0)
def e2m3(x): return qfp(x, 2, 3, 1, 7.5)

def make_mm(q, maxv, fixed):
    def scale_for(t):
        if fixed is not None: return fixed
        m = float(t.abs().max()); return 2.0 ** np.floor(np.log2(maxv / max(m, 1e-30)))
    def mm(x, w):  # x [.., K], w [N, K]
        xh, wh = f16(x), f16(w)
        xl, wl = x - xh, w - wh
        sx, sw = scale_for(x), scale_for(w)
        sxl, swl = scale_for(xl), scale_for(wl)
        y = xh @ wh.t()
        y = y + (q(xh * sx) @ q(wl * swl).t()) / (sx * swl) + (q(xl * sxl) @ q(wh * sw).t()) / (sxl * sw)
        return y
    return mm

def forward(W, mel, heads, mm):
    dt = torch.float64
    W = {k: torch.from_numpy(np.asarray(v)).to(dt) for k, v in W.items()}
    x = torch.from_numpy(np.asarray(mel)).to(dt)
    d = W["conv1.weight"].shape[0]; hd = d // heads
    h = F.gelu(F.conv1d(x, W["conv1.weight"], W["conv1.bias"], padding=1))
    h = F.gelu(F.conv1d(h, W["conv2.weight"], W["conv2.bias"], stride=2, padding=1))
    h = h.permute(0, 2, 1) + W["embed_positions.weight"]
    n_layers = 1 + max(int(k.split(".")[1]) for k in W if k.startswith("layers."))
    B, S_, _ = h.shape
    lin = (lambda x, w, b=None: (mm(x, w) if mm else x @ w.t()) + (0 if b is None else b))
    for i in range(n_layers):
        p = f"layers.{i}."
        y = F.layer_norm(h, (d,), W[p + "self_attn_layer_norm.weight"], W[p + "self_attn_layer_norm.bias"], 1e-5)
        q = (lin(y, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"]) * hd ** -0.5).view(B, S_, heads, hd).transpose(1, 2)
        k = lin(y, W[p + "self_attn.k_proj.weight"]).view(B, S_, heads, hd).transpose(1, 2)
        v = lin(y, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"]).view(B, S_, heads, hd).transpose(1, 2)
        att = (torch.softmax(q @ k.transpose(2, 3), dim=-1) @ v).transpose(1, 2).reshape(B, S_, d)
        h = h + lin(att, W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"])
        y = F.layer_norm(h, (d,), W[p + "final_layer_norm.weight"], W[p + "final_layer_norm.bias"], 1e-5)
        y = F.gelu(lin(y, W[p + "fc1.weight"], W[p + "fc1.bias"]))
        h = h + lin(y, W[p + "fc2.weight"], W[p + "fc2.bias"])
    return F.layer_norm(h, (d,), W["layer_norm.weight"], W["layer_norm.bias"], 1e-5)

model = sys.argv[1] if len(sys.argv) > 1 else "tiny"
cfg = wts.config(model)
pcm = synth.synth_clips_i16(1, seed=1234)
mel = olm.whisper_logmel(pcm.astype(np.float32) / 32768.0)
for profile, outl in [("hf", False), ("test", False), ("hf", True)]:
    W = wts.init_encoder_weights(cfg, 0, profile)
    if outl: W = wts.with_outlier_channels(W, cfg)
    ref = forward(W, mel, cfg.heads, None)
    line = f"{model} {profile} outl={outl} |ref|max {float(ref.abs().max()):.1f}:"
    for name, q, mx in [("e4m3 dyn", e4m3, 448.0), ("e3m2 dyn", e3m2, 28.0), ("e2m3 dyn", e2m3, 7.5)]:
        got = forward(W, mel, cfg.heads, make_mm(q, mx, None))
        line += f"  {name} {float((got - ref).abs().max()):.2e}"
    print(line, flush=True)
