"""GPU: BASELINE.json's full-size workload (Whisper-small, parity mode, 64 clips) checked through size-independent
properties -- the oracle needs ~1 s per clip at this size, so only a sample of clips is compared against it."""
import numpy as np
import pytest
import torch

from mlx8_ws_audio_transformer_amd import synth, weights as wts
from oracle import encoder as oracle_enc
from oracle import logmel as oracle_mel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    cfg = wts.config("small")
    pcm = torch.from_numpy(synth.synth_clips_i16(64, seed=1234, first=100)).cuda()
    enc = NativeWhisperEncoder(cfg, precision="bf16x3", seed=0, init_profile="hf").eval()
    hidden, feats = enc.encode_pcm(pcm, return_features=True)
    return cfg, pcm, enc, hidden, feats


def test_rows_are_layernormed_and_finite(full):
    cfg, pcm, enc, hidden, feats = full
    assert tuple(hidden.shape) == (64, 1500, 768) and torch.isfinite(hidden).all()
    # final LayerNorm with gamma 1 / beta 0: every row has mean 0 and variance 1
    assert hidden.mean(-1).abs().max().item() < 1e-4
    assert (hidden.var(-1, unbiased=False) - 1).abs().max().item() < 1e-3
    # log-mel: padding frames are one constant per clip, clip maximum minus 2.0 after the (x + 4) / 4 rescale
    assert torch.equal(feats[:, :, 402:], feats[:, :1, 402:403].expand(-1, 80, 2598))
    assert (feats.amax(dim=(1, 2)) - 2.0 - feats[:, 0, 402]).abs().max().item() < 1e-6


def test_batch_permutation_and_chunking_are_exact(full):
    cfg, pcm, enc, hidden, feats = full
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(5)).cuda()
    assert torch.equal(enc.encode_pcm(pcm[perm].contiguous()), hidden[perm])        # clips are independent
    small_chunks = NativeWhisperEncoder(cfg, precision="bf16x3", seed=0, init_profile="hf", chunk_clips=24).eval()
    assert torch.equal(small_chunks.encode_pcm(pcm), hidden)                            # chunk size is invisible


def test_duplicate_and_silent_clips(full):
    cfg, pcm, enc, hidden, feats = full
    p2 = pcm.clone()
    p2[7] = p2[3]
    p2[11] = 0
    h2, f2 = enc.encode_pcm(p2, return_features=True)
    assert torch.equal(h2[7], h2[3]) and torch.equal(h2[3], hidden[3])
    assert torch.all(f2[11] == -1.5)                                                    # (log10(1e-10) + 4) / 4
    assert torch.isfinite(h2[11]).all()


def test_sampled_clips_match_oracle(full):
    cfg, pcm, enc, hidden, feats = full
    idx = [0, 31, 63]
    clips = [synth.pcm_i16_to_f32(pcm[i].cpu().numpy()) for i in idx]
    mel = oracle_mel.whisper_logmel(clips)
    np.testing.assert_allclose(feats[idx].cpu().numpy(), mel, rtol=0, atol=1e-5)
    ref = oracle_enc.encoder_forward(wts.init_encoder_weights(cfg, 0, "hf"), mel, cfg.heads).numpy()
    assert np.abs(hidden[idx].cpu().numpy() - ref).max() < 1e-3
