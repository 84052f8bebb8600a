#!/usr/bin/env python3
"""Does splitting the 64-clip step into independent sub-batches on separate HIP streams help (tails of one launch filled by the other,
HBM-bound LayerNorm beside MFMA-bound GEMMs)?  GPU box only.
Measured (round 2): no -- 1 x 64 clips 51.2 ms, 2 x 32 clips 53.5 ms, 4 x 16 clips 55.6 ms per 64 clips: the GEMMs already fill every CU
and the part is power-limited, so co-scheduled launches only shorten each other's tiles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx8_ws_audio_transformer_amd import _lib, synth, weights as wts
from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder

B = 64
pcm_np = synth.synth_clips_i16(B, seed=1234)
enc = NativeWhisperEncoder(wts.config("small"), seed=0, init_profile="hf").eval()
pcm = torch.from_numpy(pcm_np).cuda()
L = _lib.lib()
enc.sync_weights()
S, d = enc.cfg.max_source_positions, enc.cfg.d_model
ref = enc.encode_pcm(pcm).clone()

def run(parts, iters=12):
    Bp = B // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    wss = [_lib.workspace(L.awt_audio_encode_workspace_bytes(enc._handle, Bp), "cuda") for _ in range(parts)]
    outs = [torch.empty((Bp, S, d), dtype=torch.float32, device="cuda") for _ in range(parts)]
    torch.cuda.synchronize()     # the buffers were allocated under the default stream: nothing queued there may still touch their memory
    def step():
        for i, st in enumerate(streams):
            p = pcm[i * Bp:(i + 1) * Bp]
            _lib.check(L.awt_audio_encode(enc._handle, _lib.ptr(p), 1, p.stride(0), None, p.shape[1], Bp, None, _lib.ptr(outs[i]), _lib.ptr(wss[i]),
                                          wss[i].numel(), st.cuda_stream))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / iters * 1e3
    err = float((torch.cat(outs) - ref).abs().max())
    print(f"{parts} stream(s) x {Bp} clips: {ms:7.2f} ms per 64 clips = {B / ms * 1e3:7.1f} clips/s   max|out - single| {err:.1e}", flush=True)

for parts in (1, 2, 4, 1, 2):
    run(parts)
