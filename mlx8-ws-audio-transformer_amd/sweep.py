"""End-to-end sweep over a pre-staged clip set (BASELINE.json configs[4], SURVEY.md §8d "C5" / §8e).

The reference evaluates its synthetic piano set one file at a time (`torchaudio.load -> processor -> model`,
/root/reference/AB/fineTuneMidiTester.py:26-36; the set comes from AB/synthDataset.py:43-91).  Here the whole shard of a
rank is ONE contiguous int16 device tensor (clip i at row i, 64 000 samples), walked in batches of `batch` clips -- the
last batch of a shard is short -- through `awt_audio_encode` (log-mel + encoder, one library call per batch).  Ranks own
disjoint contiguous shards (`dist.shard_range`); there is no data-path collective.
"""
from __future__ import annotations

from typing import Callable, Iterator, Optional, Tuple

import numpy as np
import torch

from . import synth
from .dist import shard_range


def stage_shard(n_clips: int, rank: int, world: int, seed: int = 1234, workers: int = 0) -> Tuple[np.ndarray, int]:
    """This rank's contiguous shard of the seeded `n_clips`-clip set as int16 [n_local, 64000] (host), and its first clip index."""
    lo, hi = shard_range(n_clips, rank, world)
    return synth.synth_clips_i16_parallel(hi - lo, seed=seed, first=lo, workers=workers), lo


def batches(n_local: int, batch: int) -> Iterator[Tuple[int, int]]:
    """[b0, b1) row ranges of a shard, in order; the tail batch is short."""
    for b0 in range(0, n_local, batch):
        yield b0, min(n_local, b0 + batch)


def encode_sweep(enc, pcm: torch.Tensor, batch: int = 64, sink: Optional[Callable[[int, torch.Tensor], None]] = None) -> int:
    """Walks the device shard `pcm` [n_local, n] through `enc.encode_pcm` batch by batch.  `sink(b0, hidden)` sees every
    batch's hidden states [b, S, d] (a view of a buffer that the next batch overwrites is NOT handed out: each call
    returns a fresh tensor).  Returns the number of clips encoded."""
    n = pcm.shape[0]
    done = 0
    for b0, b1 in batches(n, batch):
        hidden = enc.encode_pcm(pcm[b0:b1])
        if sink is not None:
            sink(b0, hidden)
        done += b1 - b0
    return done
