"""Soak: 400 encoder steps of the bench workload; checks that every 50th output is bit-identical to the first (GPU box only)."""
import sys, time
sys.path.insert(0, ".")
import torch
from mlx8_ws_audio_transformer_amd import weights as wts, synth
from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
cfg = wts.config("small")
enc = NativeWhisperEncoder(cfg, precision=sys.argv[1] if len(sys.argv) > 1 else None).eval()   # default: the mode chosen from the weights (f16f8)
pcm = torch.from_numpy(synth.synth_clips_i16(64, seed=1234)).cuda()
ref = enc.encode_pcm(pcm).clone()
t0 = time.time(); bad = 0
for i in range(400):
    out = enc.encode_pcm(pcm)
    if i % 50 == 49:
        ok = torch.equal(out, ref); bad += (not ok)
        print(i + 1, "steps", "bit-identical" if ok else "MISMATCH", f"{(i+1)*64/(time.time()-t0):.0f} clips/s", flush=True)
print("finite:", bool(torch.isfinite(out).all()), "mismatches:", bad)
