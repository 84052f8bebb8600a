"""Shared seeded inputs for the parity tests (regenerates exactly what tools/make_golden.py fed the reference)."""
import os

import numpy as np

from mlx8_ws_audio_transformer_amd import synth, weights as wts

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLD, name))


def logmel_inputs():
    noise = (0.1 * wts.unit_variates("f2_noise", 64000, 0)).astype(np.float32)
    tone = synth.tone_noise_clip(0)
    zeros = np.zeros(64000, dtype=np.float32)
    short = tone[:16000].copy()
    piano = synth.pcm_i16_to_f32(synth.synth_clips_i16(1, seed=1234, first=3)[0])
    return {"noise": noise, "tone": tone, "zeros": zeros, "short": short, "piano": piano}


def piano_clips_f32(batch, first=0):
    return [synth.pcm_i16_to_f32(c) for c in synth.synth_clips_i16(batch, seed=1234, first=first)]


def real_audio():
    """(int16 stereo excerpt [64000, 2] of the reference's sample recording, its channel mean as float32, the fixture): tests/golden/real_audio.npz
    (tools/make_golden.py gen_real_audio)."""
    G = golden("real_audio.npz")
    pcm = G["pcm_i16_stereo"]
    return pcm, (pcm.astype(np.float32) / 32768.0).mean(axis=1), G
