"""GPU: the reference's inference call sites (a14): `transcribe_audio_FT` (AB/wavToWhisper.py:44-70) and the tester loop
(AB/fineTuneMidiTester.py:26-49) from WAV files to text rows, over the native log-mel + encoder and the pinned greedy decoder."""
import csv
import struct

import numpy as np
import pytest
import torch

from mlx8_ws_audio_transformer_amd import synth
from tests.test_gpu_finetune import _golden_model

pytestmark = pytest.mark.gpu


def _write_wav(path, pcm_i16, rate=16000, channels=1):
    data = np.ascontiguousarray(pcm_i16, dtype="<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, channels, rate, rate * channels * 2, channels * 2, 16))
        f.write(b"data" + struct.pack("<I", len(data)) + data)


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    from mlx8_ws_audio_transformer_amd.feature_extraction import WhisperFeatureExtractor, WhisperProcessor
    from mlx8_ws_audio_transformer_amd.transcribe import NoteTokenizer
    d = tmp_path_factory.mktemp("wavs")
    model, _ = _golden_model()                                  # mini encoder (S = 200: 4 s clips) + 2-layer decoder, vocab 512, pinned to HF
    tok = NoteTokenizer(bos_token_id=1, eos_token_id=2, pad_token_id=0)
    proc = WhisperProcessor(WhisperFeatureExtractor(chunk_length=4), tokenizer=tok)      # the trimmed model's 4 s window (S = 200)
    pcm = synth.synth_clips_i16(3, seed=1234, first=7)
    paths = []
    for i in range(3):
        p = d / f"clip{i}.wav"
        _write_wav(p, pcm[i])
        paths.append(p)
    return d, model, proc, tok, pcm, paths


def _direct(model, proc, tok, wave_f32):
    feats = proc(wave_f32, sampling_rate=16000, return_tensors="pt")["input_features"].cuda()
    return tok.batch_decode(model.generate(feats, max_length=12).cpu(), skip_special_tokens=True)[0].strip()


def test_transcribe_audio_ft_writes_text_and_row(setup):
    from mlx8_ws_audio_transformer_amd.transcribe import transcribe_audio_FT
    d, model, proc, tok, pcm, paths = setup
    results = []
    text = transcribe_audio_FT(paths[0], results, model, proc, max_length=12)
    assert text == _direct(model, proc, tok, synth.pcm_i16_to_f32(pcm[0]))
    assert len(text.split()) > 0                                  # the random-init decoder emits vocabulary words, not only specials
    assert (d / "clip0.text").read_text() == f"clip0.wav: {text}\n"
    assert results == [{"Path": paths[0], "Transcription": text, "Actual": "Asmoranomardicadaistinaculdacar"}]


def test_tester_loop_over_csv_batched_equals_per_file(setup, capsys):
    from mlx8_ws_audio_transformer_amd.transcribe import evaluate_csv
    d, model, proc, tok, pcm, paths = setup
    ds = d / "mididataset.csv"
    with open(ds, "w", newline="") as f:
        w = csv.writer(f); w.writerow(["WavPath", "Labels"])
        for i, p in enumerate(paths):
            w.writerow([str(p), synth.clip_label(1234, 7 + i)])
        w.writerow([str(d / "missing.wav"), "<|MIDI|> C4 <|/MIDI|>"])
    one = evaluate_csv(ds, model, proc, batch_size=1, max_length=12)
    assert "Missing file" in capsys.readouterr().out and len(one) == 3
    out_csv = d / "midiDatasetResults.csv"
    many = evaluate_csv(ds, model, proc, out_csv=out_csv, batch_size=2, max_length=12)
    assert many == one                                             # batching the B = 1 loop does not change a prediction
    assert [r["Actual"] for r in one] == [synth.clip_label(1234, 7 + i) for i in range(3)]
    assert one[1]["Predicted"] == _direct(model, proc, tok, synth.pcm_i16_to_f32(pcm[1]))
    rows = list(csv.DictReader(open(out_csv)))
    assert rows == many and list(rows[0]) == ["WavPath", "Predicted", "Actual"]


def test_other_sample_rates_and_stereo_are_converted(setup):
    """A 32 kHz stereo file comes back as the 16 kHz mono sampling of the same signal (channel mean + libawt's resampler)."""
    from mlx8_ws_audio_transformer_amd.transcribe import load_clip_16k
    d = setup[0]
    sig = lambda t: 0.4 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 1000.0 * t + 0.3)
    t32, t16 = np.arange(64000) / 32000.0, np.arange(32000) / 16000.0
    left, right = sig(t32) + 0.1, sig(t32) - 0.1                      # the channel mean removes the offsets
    stereo = np.stack([left, right], axis=1)
    p = d / "tones_32k_stereo.wav"
    _write_wav(p, np.clip(np.rint(stereo * 32767.0), -32768, 32767).astype(np.int16), rate=32000, channels=2)
    y = load_clip_16k(p).numpy()
    assert y.shape[0] == 32000
    want = sig(t16) * (32767.0 / 32768.0)
    assert np.abs(y - want)[200:-200].max() < 2e-3                     # away from the clip edges (the sinc kernel sees zeros there)
