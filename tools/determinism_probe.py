#!/usr/bin/env python3
"""Is the encoder output independent of workspace contents, stream and call history?  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx8_ws_audio_transformer_amd import _lib, synth, weights as wts
from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else None
enc = NativeWhisperEncoder(wts.config("small"), seed=0, init_profile="hf", precision=prec).eval()
pcm = torch.from_numpy(synth.synth_clips_i16(B, seed=1234)).cuda()
L = _lib.lib()
S, d = enc.cfg.max_source_positions, enc.cfg.d_model
r1 = enc.encode_pcm(pcm).clone()
r2 = enc.encode_pcm(pcm).clone()
torch.cuda.synchronize()
print("encode_pcm twice:", float((r1 - r2).abs().max()))

def direct(fill, stream):
    ws = _lib.workspace(L.awt_audio_encode_workspace_bytes(enc._handle, B), "cuda")
    if fill is not None:
        ws.fill_(fill)
    out = torch.empty((B, S, d), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    _lib.check(L.awt_audio_encode(enc._handle, _lib.ptr(pcm), 1, pcm.stride(0), None, pcm.shape[1], B, None, _lib.ptr(out), _lib.ptr(ws), ws.numel(), stream))
    torch.cuda.synchronize()
    return out

for fill in (None, 0, 0xFF, 0x3C, 0x7B):
    o = direct(fill, torch.cuda.current_stream().cuda_stream)
    print(f"fresh workspace fill {fill}: max diff {float((o - r1).abs().max()):.3e}  nan {bool(torch.isnan(o).any())}", flush=True)
st = torch.cuda.Stream()
o = direct(0, st.cuda_stream)
print(f"side stream, zero workspace: {float((o - r1).abs().max()):.3e}")
for i in range(3):
    o = enc.encode_pcm(pcm)
    print("encode_pcm again:", float((o - r1).abs().max()))
