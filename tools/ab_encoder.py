#!/usr/bin/env python3
"""A/B of libawt builds on the headline workload, interleaved in ONE process on ONE device (cdna_hip_programming.md section 5.4 rule 24).

    python tools/ab_encoder.py [--model small] [--batch 64] [--weights fp16,fp32] [--rounds 5] [--steps 3] [--precision f16f8] lib0.so lib1.so ...

Every library is loaded through its own ctypes handle (distinct files -> distinct images), gets its own awt_ctx and encoder with the
bench's deterministic weights, and runs `awt_audio_encode` on the same resident int16 clips; rounds visit the libraries in turn.  Per
library: median / min ms per step (torch events), the GEMM / attention / LayerNorm class times of the library's own event timers, and
the largest difference of its hidden states from the first library's (0 = bit-identical).  GPU box only.
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from mlx8_ws_audio_transformer_amd import _lib, synth, weights as wts
from mlx8_ws_audio_transformer_amd.encoder import PRECISIONS


class Lib:
    def __init__(self, spec):
        # "path.so" or "path.so:key=value,key=value" (awt_tuning_set on that library image; the SAME file may not be listed twice: one image)
        path, _, tun = spec.partition(":")
        self.path = spec
        self.L = C.CDLL(os.path.abspath(path))
        for name, (res, args) in _lib._SIGNATURES.items():
            fn = getattr(self.L, name)
            fn.restype, fn.argtypes = res, args
        out = C.c_void_p()
        self.check(self.L.awt_ctx_create(0, C.byref(out)))
        self.ctx = out.value
        self.enc = {}
        for kv in filter(None, tun.split(",")):
            k, v = kv.split("=")
            self.check(self.L.awt_tuning_set(k.encode(), int(v)))

    def check(self, rc):
        if rc != 0:
            raise RuntimeError("%s: %s" % (self.path, self.L.awt_last_error().decode()))

    def encoder(self, key, cfg, precision, W):
        ecfg = _lib.EncoderCfg(cfg.d_model, cfg.layers, cfg.heads, cfg.ffn, cfg.n_mels, cfg.max_source_positions, PRECISIONS[precision], 0, 0.0, 0, 0, 0, 0)
        out = C.c_void_p()
        self.check(self.L.awt_encoder_create(self.ctx, C.byref(ecfg), C.byref(out)))
        for name, arr in W.items():
            t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).cuda()
            shape = (C.c_int64 * t.dim())(*t.shape)
            self.check(self.L.awt_encoder_set_weight(out.value, name.encode(), t.data_ptr(), shape, t.dim(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        self.enc[key] = out.value
        return out.value

    def encode(self, key, pcm, hidden, ws):
        B, n = pcm.shape
        self.check(self.L.awt_audio_encode(self.enc[key], pcm.data_ptr(), 1, pcm.stride(0), None, n, B, None, hidden.data_ptr(), ws.data_ptr(), ws.numel(),
                                           torch.cuda.current_stream().cuda_stream))

    def prof(self, on):
        mask = 0
        if on:
            for k in ("gemm", "attention", "layernorm", "logmel"):
                mask |= 1 << _lib.PROF_CLASSES[k]
        self.check(self.L.awt_prof_enable(self.ctx, mask))

    def collect(self, klass):
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        self.check(self.L.awt_prof_collect(self.ctx, _lib.PROF_CLASSES[klass], C.byref(ms), C.byref(n), C.byref(fl)))
        return ms.value, n.value, fl.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--model", default="small")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--weights", default="fp32,fp16")
    ap.add_argument("--precision", default="f16f8")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    cfg = wts.config(a.model, False)
    pcm = torch.from_numpy(synth.synth_clips_i16(a.batch, seed=1234, first=0)).cuda()
    libs = [Lib(p) for p in a.libs]
    hidden = torch.empty((a.batch, cfg.max_source_positions, cfg.d_model), dtype=torch.float32, device="cuda")
    for kind in a.weights.split(","):
        W = bench.bench_weights(cfg, kind)
        ws_bytes = 0
        for lb in libs:
            h = lb.encoder(kind, cfg, a.precision, W)
            ws_bytes = max(ws_bytes, lb.L.awt_audio_encode_workspace_bytes(h, a.batch))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
        ref = None
        diffs = []
        for lb in libs:                                   # warm-up + output comparison
            lb.encode(kind, pcm, hidden, ws); lb.encode(kind, pcm, hidden, ws)
            torch.cuda.synchronize()
            if ref is None:
                ref = hidden.clone()
            diffs.append(float((hidden - ref).abs().max()))
        times = [[] for _ in libs]
        klass = [dict(gemm=[], attention=[], layernorm=[]) for _ in libs]
        for r in range(a.rounds):
            for i, lb in enumerate(libs):
                lb.prof(True)
                for k in klass[i]:
                    lb.collect(k)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.steps):
                    lb.encode(kind, pcm, hidden, ws)
                e1.record(); torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / a.steps)
                for k in klass[i]:
                    klass[i][k].append(lb.collect(k)[0] / a.steps)
                lb.prof(False)
        med = lambda v: sorted(v)[len(v) // 2]
        print("== Whisper-%s B=%d %s, weights %s (%d rounds x %d steps, interleaved)" % (a.model, a.batch, a.precision, kind, a.rounds, a.steps))
        for i, lb in enumerate(libs):
            print("  %-60s step %7.2f ms (min %7.2f)  %7.1f clips/s | gemm %6.2f attn %6.2f ln %5.2f | max|h - h(lib0)| %.1e" % (
                os.path.basename(lb.path), med(times[i]), min(times[i]), a.batch / med(times[i]) * 1e3, med(klass[i]["gemm"]), med(klass[i]["attention"]),
                med(klass[i]["layernorm"]), diffs[i]), flush=True)
        del ws


if __name__ == "__main__":
    main()
