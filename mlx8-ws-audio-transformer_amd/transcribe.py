"""The reference's inference call sites over the native path: WAV file -> log-mel -> encoder -> greedy decode -> text.

Mirrors
* `transcribe_audio_FT(input_path, results)`     /root/reference/AB/wavToWhisper.py:44-70
      torchaudio.load -> processor(waveform, sampling_rate=16000, return_tensors="pt") -> model.generate(input_features)
      -> processor.batch_decode(ids, skip_special_tokens=True)[0].strip() -> `<stem>.text` + a results row
  (the reference reloads the checkpoint from disk for every file, wavToWhisper.py:47; here model and processor are arguments);
* the tester loop                                  /root/reference/AB/fineTuneMidiTester.py:26-49
      rows of `mididataset.csv` (WavPath, Labels) -> the same chain -> rows {WavPath, Predicted, Actual} -> CSV.

`torchaudio.load` is replaced by `urbansound.read_wav` (RIFF/WAVE PCM16 / float32), the channel mix and -- when the file is not at
16 kHz -- the sample-rate conversion run in libawt (`awt_prepare_waveform`); the reference passes whatever rate the file has as
"16000", which only works for its own 16 kHz recordings (AB/memoToWav.py:16-21 writes 16 kHz mono s16).

Tokenizer files are not available offline and tokenisation is outside the hot path; `NoteTokenizer` is a small
stand-in with the Whisper tokenizer's call surface for the synthetic piano set's labels (`<|MIDI|> G#6 F2 ... <|/MIDI|>`,
AB/synthDataset.py:48-49,73-74,82) so that the whole chain can be exercised end to end.  Any object with `__call__(text)["input_ids"]`,
`batch_decode` and `pad` (e.g. a real `WhisperTokenizer`) plugs into `WhisperProcessor(tokenizer=...)` the same way.
"""
from __future__ import annotations

import csv
import os
from pathlib import Path
from typing import Any, Dict, Iterable, List, Optional, Sequence, Union

import torch

from .feature_extraction import WhisperProcessor
from .synth import NOTE_NAMES


class NoteTokenizer:
    """Word-level tokenizer over `<|MIDI|>`, `<|/MIDI|>` and the 128 MIDI note names, with Whisper's special-token layout
    (`<|startoftranscript|>` first, `<|endoftext|>` last and as padding)."""

    def __init__(self, bos_token_id: int = 1, eos_token_id: int = 2, pad_token_id: Optional[int] = None, first_id: int = 3):
        self.bos_token_id, self.eos_token_id = bos_token_id, eos_token_id
        self.pad_token_id = eos_token_id if pad_token_id is None else pad_token_id
        words = ["<|MIDI|>", "<|/MIDI|>"] + [f"{NOTE_NAMES[n % 12]}{n // 12 - 1}" for n in range(128)]
        self.vocab = {w: first_id + i for i, w in enumerate(words)}
        self.unk_token_id = first_id + len(words)
        self.inverse = {i: w for w, i in self.vocab.items()}
        self.special = {self.bos_token_id, self.eos_token_id, self.pad_token_id}
        self.vocab_size = self.unk_token_id + 1

    def __call__(self, text: Union[str, Sequence[str]], **kwargs) -> Dict[str, Any]:
        def one(t: str) -> List[int]:
            return [self.bos_token_id] + [self.vocab.get(w, self.unk_token_id) for w in t.split()] + [self.eos_token_id]
        if isinstance(text, str):
            return {"input_ids": one(text)}
        return {"input_ids": [one(t) for t in text]}

    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        out = []
        for i in (ids.tolist() if hasattr(ids, "tolist") else list(ids)):
            i = int(i)
            if i < 0 or (skip_special_tokens and i in self.special):
                continue
            out.append(self.inverse.get(i, "<|startoftranscript|>" if i == self.bos_token_id else "<|endoftext|>" if i in (self.eos_token_id, self.pad_token_id) else "<unk>"))
        return " ".join(out)

    def batch_decode(self, sequences, skip_special_tokens: bool = False) -> List[str]:
        return [self.decode(s, skip_special_tokens=skip_special_tokens) for s in sequences]

    def pad(self, features: Sequence[Dict[str, Sequence[int]]], return_tensors: str = "pt") -> Dict[str, torch.Tensor]:
        width = max(len(f["input_ids"]) for f in features)
        ids = torch.full((len(features), width), self.pad_token_id, dtype=torch.long)
        mask = torch.zeros((len(features), width), dtype=torch.long)
        for r, f in enumerate(features):
            n = len(f["input_ids"])
            ids[r, :n] = torch.as_tensor(list(f["input_ids"]), dtype=torch.long)
            mask[r, :n] = 1
        return {"input_ids": ids, "attention_mask": mask}


def load_clip_16k(path: Union[str, os.PathLike]) -> torch.Tensor:
    """WAV file -> mono float32 waveform at 16 kHz on the CPU.  16 kHz files take the loader's convention (int16 / 32768, channel
    mean); any other rate goes through libawt's resampler (`urbansound.prepare_waveform` -> `awt_prepare_waveform`)."""
    from .urbansound import prepare_waveform, read_wav, resampled_length
    samples, sr = read_wav(str(path))                        # [n, C] in file order
    if sr == 16000:
        x = samples.float() / 32768.0 if samples.dtype == torch.int16 else samples.float()
        return x.mean(dim=1) if x.shape[1] > 1 else x[:, 0].contiguous()
    full = prepare_waveform(samples, sample_rate=sr, target_rate=16000, interleaved=True, n_out=resampled_length(samples.shape[0], sr, 16000))
    return full[0].cpu()


@torch.no_grad()
def transcribe_clips(waveforms: Sequence[Any], model, processor: WhisperProcessor, max_length: int = 225) -> List[str]:
    """A batch of mono 16 kHz waveforms -> stripped transcriptions: `processor` -> `model.generate` -> `processor.batch_decode`."""
    inputs = processor([w.numpy() if hasattr(w, "numpy") else w for w in waveforms], sampling_rate=16000, return_tensors="pt")
    dev = model.encoder.device
    generated_ids = model.generate(inputs["input_features"].to(dev), max_length=max_length)
    return [t.strip() for t in processor.batch_decode(generated_ids.cpu(), skip_special_tokens=True)]


def transcribe_audio_FT(input_path, results: List[dict], model, processor: WhisperProcessor, actual: str = "Asmoranomardicadaistinaculdacar",
                        write_text: bool = True, max_length: int = 225) -> str:
    """wavToWhisper.py:44-70 with the model and processor passed in: transcribes one file, writes `<stem>.text` and appends the
    reference's row {"Path", "Transcription", "Actual"} to `results`."""
    input_path = Path(input_path)
    transcription = transcribe_clips([load_clip_16k(input_path)], model, processor, max_length)[0]
    if write_text:
        with open(input_path.with_suffix(".text"), "w") as f:
            f.write(f"{input_path.name}: {transcription}\n")
    results.append({"Path": input_path, "Transcription": transcription, "Actual": actual})
    return transcription


def evaluate_csv(dataset_csv, model, processor: WhisperProcessor, out_csv=None, batch_size: int = 16, max_length: int = 225,
                 root: Optional[str] = None) -> List[Dict[str, str]]:
    """fineTuneMidiTester.py:16-49: every row (WavPath, Labels) of `dataset_csv` -> {"WavPath", "Predicted", "Actual"}; missing files are
    reported and skipped like the reference does.  Clips go through the model `batch_size` at a time (the reference's B = 1 loop
    is `batch_size=1`); `out_csv` writes `midiDatasetResults.csv`."""
    rows = list(csv.DictReader(open(dataset_csv, newline="")))
    todo = []
    for row in rows:
        wav_path = Path(row["WavPath"]) if root is None else Path(root) / row["WavPath"]
        if not wav_path.exists():
            print(f"Missing file: {wav_path}")
            continue
        todo.append((wav_path, row["Labels"]))
    midi_results: List[Dict[str, str]] = []
    for i in range(0, len(todo), max(1, batch_size)):
        chunk = todo[i: i + max(1, batch_size)]
        texts = transcribe_clips([load_clip_16k(p) for p, _ in chunk], model, processor, max_length)
        for (p, actual), pred in zip(chunk, texts):
            midi_results.append({"WavPath": str(p), "Predicted": pred, "Actual": actual})
    if out_csv is not None:
        with open(out_csv, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["WavPath", "Predicted", "Actual"])
            w.writeheader()
            w.writerows(midi_results)
    return midi_results
