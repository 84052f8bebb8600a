"""The stock-PyTorch decoder of the fine-tune step (finetune.WhisperDecoder, scope row "next" #1) pinned to the
reference's WhisperForConditionalGeneration through tests/golden/decoder.npz; plus the WER helper.  CPU only."""
import numpy as np
import torch
import torch.nn.functional as F

from mlx8_ws_audio_transformer_amd import weights as wts
from mlx8_ws_audio_transformer_amd.finetune import WhisperDecoder, greedy_decode, shift_tokens_right, wer
from oracle import encoder as oracle_enc, logmel as oracle_mel
from tests.util import golden, piano_clips_f32

G = golden("decoder.npz")


def _setup():
    cfg = wts.config("mini", True)
    We = wts.init_encoder_weights(cfg, seed=0, profile="test")
    mel = oracle_mel.whisper_logmel(piano_clips_f32(2), n_samples=2 * cfg.max_source_positions * 160)
    hidden = oracle_enc.encoder_forward(We, mel, cfg.heads)
    dec = WhisperDecoder(cfg.d_model, 2, cfg.heads, cfg.ffn, vocab=512, max_target_positions=64).eval()
    Wd = wts.init_decoder_weights(cfg.d_model, 2, cfg.ffn, 512, 64, seed=0)
    assert [k for k, _ in wts.decoder_param_shapes(cfg.d_model, 2, cfg.ffn, 512, 64)] != []
    assert set(dec.state_dict()) == set(Wd)                       # HF WhisperDecoder.state_dict() keys
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in Wd.items()}, strict=True)
    return dec, hidden


def test_logits_loss_and_greedy_tokens_match_reference():
    dec, hidden = _setup()
    np.testing.assert_allclose(hidden[:, :4].numpy(), G["encoder_head"], rtol=0, atol=2e-4)
    labels = torch.from_numpy(G["labels"])
    ids = shift_tokens_right(labels, pad_token_id=0, decoder_start_token_id=1)
    with torch.no_grad():
        logits = dec(ids, hidden)
    np.testing.assert_allclose(logits.numpy(), G["logits"], rtol=0, atol=5e-4)
    loss = F.cross_entropy(logits.view(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=-100)
    assert abs(float(loss) - float(G["loss"])) < 1e-4
    out = greedy_decode(dec, hidden, start_id=1, pad_id=0, eos_id=2, max_length=G["greedy_ids"].shape[1])
    np.testing.assert_array_equal(out.numpy(), G["greedy_ids"])     # incremental (cached) decoding == full re-evaluation


def test_shift_tokens_right_rules():
    lab = torch.tensor([[5, 6, -100, -100], [7, 8, 9, 10]])
    np.testing.assert_array_equal(shift_tokens_right(lab, 0, 1).numpy(), [[1, 5, 6, 0], [1, 7, 8, 9]])


def test_wer():
    assert wer(["a b c d"], ["a b c d"]) == 0.0
    assert wer(["a b c d"], ["a x c"]) == 0.5                       # one substitution + one deletion
    assert wer(["a b"], ["a b c d"]) == 1.0                         # two insertions / two reference words
    assert abs(wer(["a b c d", "hello world"], ["a x c", "hello world"]) - 2 / 6) < 1e-12
    assert wer([""], ["a"]) == 1.0
