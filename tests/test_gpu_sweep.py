"""GPU: BASELINE.json configs[4], the pre-staged clip-set sweep (sweep.py): batching, the short tail batch and the per-rank
shards are invisible in the results, and sampled clips (incl. one of the tail batch) match the oracle."""
import numpy as np
import pytest
import torch

from mlx8_ws_audio_transformer_amd import sweep, synth, weights as wts
from mlx8_ws_audio_transformer_amd.dist import shard_range
from oracle import encoder as oracle_enc
from oracle import logmel as oracle_mel

pytestmark = pytest.mark.gpu
N = 200


@pytest.fixture(scope="module")
def swept():
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    cfg = wts.config("tiny")
    host, first = sweep.stage_shard(N, 0, 1, seed=1234)
    assert first == 0 and host.shape == (N, 64000) and host.dtype == np.int16
    pcm = torch.from_numpy(host).cuda()
    enc = NativeWhisperEncoder(cfg, seed=0, init_profile="hf").eval()          # default precision (f16f8)
    out = torch.empty((N, cfg.max_source_positions, cfg.d_model), device="cuda")
    seen = []

    def sink(b0, hidden):
        seen.append((b0, hidden.shape[0]))
        out[b0: b0 + hidden.shape[0]] = hidden

    assert sweep.encode_sweep(enc, pcm, 64, sink) == N
    assert seen == [(0, 64), (64, 64), (128, 64), (192, 8)]           # three full batches and the tail batch
    return cfg, host, pcm, enc, out


def test_sweep_equals_per_batch_encode_and_other_batchings(swept):
    cfg, host, pcm, enc, out = swept
    for b0, b1 in sweep.batches(N, 64):
        assert torch.equal(enc.encode_pcm(pcm[b0:b1]), out[b0:b1])      # the sweep IS the per-batch library call
    # a different batch size changes tile shapes (fp32 summation order), never the result beyond rounding noise
    other = torch.empty_like(out)
    sweep.encode_sweep(enc, pcm, 48, lambda b0, h: other[b0: b0 + h.shape[0]].copy_(h))
    assert float((other - out).abs().max()) < 1e-4


def test_two_rank_shards_cover_the_sweep(swept):
    cfg, host, pcm, enc, out = swept
    parts = []
    for rank in range(2):
        shard, first = sweep.stage_shard(N, rank, 2, seed=1234)
        lo, hi = shard_range(N, rank, 2)
        assert first == lo and np.array_equal(shard, host[lo:hi])        # contiguous shards of the same seeded set
        got = torch.empty((hi - lo, cfg.max_source_positions, cfg.d_model), device="cuda")
        sweep.encode_sweep(enc, torch.from_numpy(shard).cuda(), 64, lambda b0, h: got[b0: b0 + h.shape[0]].copy_(h))
        parts.append(got)
    assert float((torch.cat(parts) - out).abs().max()) < 1e-4


def test_sampled_clips_match_oracle(swept):
    cfg, host, pcm, enc, out = swept
    idx = [0, 100, 199]                                                   # 199 sits in the 8-clip tail batch
    mel = oracle_mel.whisper_logmel([synth.pcm_i16_to_f32(host[i]) for i in idx])
    ref = oracle_enc.encoder_forward(wts.init_encoder_weights(cfg, 0, "hf"), mel, cfg.heads).numpy()
    assert np.abs(out[idx].cpu().numpy() - ref).max() < 1e-3
