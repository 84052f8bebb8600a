"""CPU: host side of the clip-set sweep (BASELINE.json configs[4]) -- shard staging and batch walking."""
import numpy as np

from mlx8_ws_audio_transformer_amd import sweep, synth
from mlx8_ws_audio_transformer_amd.dist import shard_range


def test_parallel_synthesis_equals_serial():
    a = synth.synth_clips_i16_parallel(300, seed=1234, first=17, workers=4)      # forks a pool (never after GPU init)
    b = synth.synth_clips_i16(300, seed=1234, first=17)
    assert a.dtype == np.int16 and np.array_equal(a, b)


def test_shards_are_contiguous_pieces_of_one_seeded_set():
    whole, first = sweep.stage_shard(21, 0, 1, seed=5)
    assert first == 0 and whole.shape == (21, 64000)
    got = []
    for r in range(4):
        part, lo = sweep.stage_shard(21, r, 4, seed=5)
        assert (lo, lo + part.shape[0]) == shard_range(21, r, 4)
        got.append(part)
    assert np.array_equal(np.concatenate(got), whole)


def test_batches_include_the_short_tail():
    assert list(sweep.batches(10000, 64))[-1] == (9984, 10000) and len(list(sweep.batches(10000, 64))) == 157
    assert list(sweep.batches(1250, 64))[-1] == (1216, 1250)                     # one rank of eight
    assert list(sweep.batches(64, 64)) == [(0, 64)] and list(sweep.batches(0, 64)) == []
