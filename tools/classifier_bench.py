#!/usr/bin/env python3
"""Throughput of the UrbanSound Transformer classifier (/root/reference/.charles/spectrogram.py:944-1164) on the native
operators: clips/s in eval() and per training step (forward + backward + Adam), for the reference's two hop lengths (hop 512 -> 126 frames, hop 128 ->
501 frames of a 4 s clip at 16 kHz), batch 32 as `train_transformer`'s loader.  GPU box only.

    python tools/classifier_bench.py [--batch 32] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mlx8_ws_audio_transformer_amd.urbansound_classifier import TransformerUrbanSound8KClassifier, native_cross_entropy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    out = {}
    for hop, T in ((512, 126), (128, 501)):
        torch.manual_seed(0)
        model = TransformerUrbanSound8KClassifier(n_mels=64).cuda()
        x = torch.randn(a.batch, 64, T, device="cuda") * 2 - 4
        y = torch.randint(0, 10, (a.batch,), device="cuda")
        model.eval()
        for _ in range(3):
            model(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            model(x)
        torch.cuda.synchronize(); ev = (time.perf_counter() - t0) / a.steps
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)

        def step():
            opt.zero_grad()
            loss = native_cross_entropy(model(x), y)
            loss.backward()
            opt.step()
            return loss
        for _ in range(3):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = step()
        torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / a.steps
        out["hop%d" % hop] = {"frames": T, "batch": a.batch, "eval_clips_per_s": round(a.batch / ev, 1), "eval_ms_per_batch": round(ev * 1e3, 3),
                              "train_clips_per_s": round(a.batch / tr, 1), "train_ms_per_step": round(tr * 1e3, 3), "last_loss": float(loss.detach()),
                              "train_dropout": 0.1}
    print(json.dumps({"metric": "UrbanSound Transformer classifier, clips/s (64 mel bins, d 128, 2 layers, 4 heads; eval: packed weights cached; train: "
                                "every operator's forward and backward on libawt, attention-probability dropout in-kernel)", "results": out}))


if __name__ == "__main__":
    main()
