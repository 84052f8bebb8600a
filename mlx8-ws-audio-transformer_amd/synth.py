"""Seeded PCM synthesiser for the throughput sweep (BASELINE.json configs[4], SURVEY.md §8d "C5").

The reference's generator (/root/reference/AB/synthDataset.py:43-91) draws, per clip, five piano
notes -- pitch ~ U{21..108}, duration ~ U{0.1,0.12,0.15,0.2,0.23,0.3} s, a gap ~ U{0.1..0.5 step 0.05} s
before and after every note -- writes them as MIDI and renders them with FluidSynth + a SoundFont.
Neither FluidSynth nor the SoundFont renderer exists on the GPU box, so only the *distributions* are
re-created here and each note is rendered as a decaying harmonic stack.  Labels keep the reference's
format (`<|MIDI|> G#6 F2 ... <|/MIDI|>`, synthDataset.py:48-49,73-74,82).

All random choices come from the integer hash in weights.py, so clip i of seed s has the same notes on
every host.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from .weights import _fnv1a64, _splitmix64, _MASK, _SEEDMIX

SAMPLE_RATE = 16000
CLIP_SAMPLES = 64000  # 4 s
NOTES = list(range(21, 109))                                  # synthDataset.py:46
DURATIONS = [0.1, 0.12, 0.15, 0.2, 0.23, 0.3]                 # synthDataset.py:50
GAPS = [0.1, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4, 0.45, 0.5]      # synthDataset.py:51
NOTE_NAMES = ["C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"]  # synthDataset.py:15-16


def note_number_to_name(n: int) -> str:
    """synthDataset.py:18-21."""
    return f"{NOTE_NAMES[n % 12]}{(n // 12) - 1}"


def _draws(seed: int, clip_index: int, count: int) -> np.ndarray:
    key = (_fnv1a64(f"clip{clip_index}") ^ ((seed * _SEEDMIX) & _MASK)) & _MASK
    with np.errstate(over="ignore"):
        ctr = np.uint64(key) + np.arange(count, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    return _splitmix64(ctr) >> np.uint64(11)


def clip_notes(seed: int, clip_index: int) -> List[Tuple[float, float, int]]:
    """[(start_s, duration_s, midi_pitch)] x5 following the draw order of synthDataset.py:64-76."""
    z = _draws(seed, clip_index, 20)
    t = 0.0
    notes = []
    for n in range(5):
        t += GAPS[int(z[4 * n] % len(GAPS))]
        pitch = NOTES[int(z[4 * n + 1] % len(NOTES))]
        dur = DURATIONS[int(z[4 * n + 2] % len(DURATIONS))]
        notes.append((t, dur, pitch))
        t += dur + GAPS[int(z[4 * n + 3] % len(GAPS))]
    return notes


def clip_label(seed: int, clip_index: int) -> str:
    return "<|MIDI|> " + " ".join(note_number_to_name(p) for _, _, p in clip_notes(seed, clip_index)) + " <|/MIDI|>"


def render_clip(seed: int, clip_index: int, n_samples: int = CLIP_SAMPLES) -> np.ndarray:
    """float64 waveform in [-1, 1]: per note  0.5 * sum_{k<=6} k^-1.5 sin(2 pi k f t) e^{-4 t}, f = 440 * 2^((p-69)/12)."""
    out = np.zeros(n_samples, dtype=np.float64)
    for start, dur, pitch in clip_notes(seed, clip_index):
        f = 440.0 * 2.0 ** ((pitch - 69) / 12.0)
        i0 = int(round(start * SAMPLE_RATE))
        n = min(int(round(dur * SAMPLE_RATE)), n_samples - i0)
        if n <= 0:
            continue
        t = np.arange(n) / SAMPLE_RATE
        env = np.exp(-4.0 * t)
        tone = np.zeros(n)
        for k in range(1, 7):
            if k * f < SAMPLE_RATE / 2:
                tone += k ** -1.5 * np.sin(2 * np.pi * k * f * t)
        out[i0: i0 + n] += 0.5 * tone * env / 1.7  # 1.7 ~ sum k^-1.5: keep the stack inside [-0.5, 0.5]
    return out


def synth_clips_i16(count: int, seed: int = 1234, first: int = 0, n_samples: int = CLIP_SAMPLES) -> np.ndarray:
    """int16 [count, n_samples] -- clips `first` .. `first + count - 1` of the seeded set (16-bit PCM like the
    reference's WAVs: /root/reference/AB/memoToWav.py:16-21 writes pcm_s16le)."""
    pcm = np.empty((count, n_samples), dtype=np.int16)
    for i in range(count):
        w = render_clip(seed, first + i, n_samples)
        pcm[i] = np.clip(np.rint(w * 32767.0), -32768, 32767).astype(np.int16)
    return pcm


def _synth_span(args) -> np.ndarray:
    count, seed, first, n_samples = args
    return synth_clips_i16(count, seed, first, n_samples)


def synth_clips_i16_parallel(count: int, seed: int = 1234, first: int = 0, n_samples: int = CLIP_SAMPLES, workers: int = 0) -> np.ndarray:
    """`synth_clips_i16` over a pool of forked host processes (2.5 ms per clip on one core: the 10 000-clip sweep set
    would take 25 s serially).  Call it BEFORE the process touches the GPU: the pool forks.  workers = 0: one per
    available core, at most 16."""
    import multiprocessing as mp
    import os

    if workers <= 0:
        workers = min(16, len(os.sched_getaffinity(0)))
    if count < 256 or workers == 1:
        return synth_clips_i16(count, seed, first, n_samples)
    per = -(-count // (4 * workers))
    spans = [(min(per, count - lo), seed, first + lo, n_samples) for lo in range(0, count, per)]
    with mp.get_context("fork").Pool(workers) as pool:
        parts = pool.map(_synth_span, spans)
    return np.concatenate(parts, axis=0)


def pcm_i16_to_f32(pcm: np.ndarray) -> np.ndarray:
    """The decode convention of the reference's loaders (soundfile / torchaudio.load): int16 / 32768."""
    return pcm.astype(np.float32) / np.float32(32768.0)


def tone_noise_clip(seed: int = 0, n_samples: int = CLIP_SAMPLES, tone_hz: float = 440.0) -> np.ndarray:
    """SURVEY.md §8d "C1" input: 0.3 sin(2 pi 440 t) + 0.05 N(0,1)-like noise, fp32."""
    from .weights import unit_variates

    t = np.arange(n_samples) / SAMPLE_RATE
    return (0.3 * np.sin(2 * np.pi * tone_hz * t) + 0.05 * unit_variates("tone_noise", n_samples, seed)).astype(np.float32)
