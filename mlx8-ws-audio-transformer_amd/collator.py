"""Host-side mirror of the reference's Dataset map function and batch collator.

Reference interface (names, argument meaning and behaviour kept so `fineTune.py` call sites read the same):

* `prepare_dataset(batch)`                      /root/reference/AB/fineTune.py:85-92
* `DataCollatorSpeechSeq2SeqWithPadding`        /root/reference/AB/fineTune.py:99-118
                                                (= /root/reference/AB/exampleDataCollator.py:5-31)

This is plumbing: list-of-dicts in, dict-of-tensors out, no arithmetic.  Label padding follows
`tokenizer.pad` (right-pad with the pad id, attention mask 1/0), masked positions become -100, and the
leading decoder-start token is dropped only if EVERY row starts with it (fineTune.py:112-115).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np
import torch


def pad_labels(label_rows: Sequence[Sequence[int]], pad_token_id: int) -> Dict[str, torch.Tensor]:
    """What `processor.tokenizer.pad([{"input_ids": ...}], return_tensors="pt")` returns for right padding."""
    width = max(len(r) for r in label_rows)
    ids = torch.full((len(label_rows), width), int(pad_token_id), dtype=torch.int64)
    mask = torch.zeros((len(label_rows), width), dtype=torch.int64)
    for i, r in enumerate(label_rows):
        ids[i, : len(r)] = torch.as_tensor(list(r), dtype=torch.int64)
        mask[i, : len(r)] = 1
    return {"input_ids": ids, "attention_mask": mask}


def stack_input_features(rows: Sequence[Any]) -> torch.Tensor:
    """`feature_extractor.pad(..., return_tensors="pt")` for Whisper features: every row is already
    [80, 3000], so padding is a pure stack to fp32 [B, 80, 3000] (nested lists and
    ndarrays give bit-identical results)."""
    arrs = []
    for r in rows:
        if isinstance(r, torch.Tensor):
            arrs.append(r.detach().to("cpu", torch.float32))
        else:
            arrs.append(torch.from_numpy(np.asarray(r, dtype=np.float32)))
    shape = arrs[0].shape
    for a in arrs:
        if a.shape != shape:
            raise ValueError(f"input_features rows differ in shape: {tuple(a.shape)} vs {tuple(shape)}")
    return torch.stack(arrs, dim=0)


@dataclass
class DataCollatorSpeechSeq2SeqWithPadding:
    """Same dataclass fields and `__call__` contract as the reference's collator.

    `processor` may be a HF `WhisperProcessor` (then its own `.feature_extractor.pad` / `.tokenizer.pad`
    are used, exactly as the reference does) or this package's `WhisperProcessor` stand-in, whose
    `.tokenizer` may be None -- in that case labels are padded with `pad_token_id` (Whisper: 50257)."""

    processor: Any
    decoder_start_token_id: int
    pad_token_id: int = 50257

    def __call__(self, features: List[Dict[str, Union[List[int], torch.Tensor]]]) -> Dict[str, torch.Tensor]:
        input_features = [{"input_features": f["input_features"]} for f in features]
        fe = getattr(self.processor, "feature_extractor", None)
        if fe is not None and hasattr(fe, "pad"):
            batch = fe.pad(input_features, return_tensors="pt")
        else:
            batch = {"input_features": stack_input_features([f["input_features"] for f in input_features])}

        label_features = [{"input_ids": f["labels"]} for f in features]
        tok = getattr(self.processor, "tokenizer", None)
        if tok is not None and hasattr(tok, "pad"):
            labels_batch = tok.pad(label_features, return_tensors="pt")
            ids, mask = labels_batch["input_ids"], labels_batch["attention_mask"]
        else:
            lb = pad_labels([f["input_ids"] for f in label_features], self.pad_token_id)
            ids, mask = lb["input_ids"], lb["attention_mask"]

        labels = ids.masked_fill(mask.ne(1), -100)
        if (labels[:, 0] == self.decoder_start_token_id).all().cpu().item():
            labels = labels[:, 1:]
        batch["labels"] = labels
        return batch


def make_prepare_dataset(processor):
    """Returns the reference's `prepare_dataset(batch)` bound to `processor` (fineTune.py:85-92)."""

    def prepare_dataset(batch):
        audio = batch["audio"]
        result = processor(audio["array"], sampling_rate=audio["sampling_rate"], text=batch["sentence"])
        batch["input_features"] = result["input_features"][0]
        batch["labels"] = result["labels"]
        return batch

    return prepare_dataset
