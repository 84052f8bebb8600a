"""GPU: the fineTune.py step over the native encoder -- loss goes down, sharded gradients average to the full-batch
gradient (the DP invariant of SURVEY.md §8c "not pinned (4)"), adapter-only checkpoint round-trips."""
import numpy as np
import pytest
import torch

from mlx8_ws_audio_transformer_amd import weights as wts
from oracle import logmel as oracle_mel
from tests.util import piano_clips_f32

pytestmark = pytest.mark.gpu


def _batch(cfg, B, first=0):
    mel = oracle_mel.whisper_logmel(piano_clips_f32(B, first), n_samples=cfg.n_frames * 160)
    g = torch.Generator().manual_seed(first)
    labels = torch.randint(0, 1000, (B, 6), generator=g)
    labels[:, 0] = 50258
    labels[0, 4:] = -100
    return {"input_features": torch.from_numpy(mel), "labels": labels}


def _model(cfg):
    from mlx8_ws_audio_transformer_amd.finetune import WhisperLoRAModel
    return WhisperLoRAModel(cfg, wts.LoraSpec(r=8, alpha=16.0), decoder_layers=1)


def test_loss_decreases_and_checkpoint_roundtrip(tmp_path):
    from mlx8_ws_audio_transformer_amd.collator import DataCollatorSpeechSeq2SeqWithPadding
    from mlx8_ws_audio_transformer_amd.finetune import Seq2SeqTrainer, Seq2SeqTrainingArguments
    cfg = wts.config("mini", True)
    model = _model(cfg)
    b = _batch(cfg, 4)
    ds = [{"input_features": b["input_features"][i].numpy(), "labels": [t for t in b["labels"][i].tolist() if t != -100]} for i in range(4)]
    args = Seq2SeqTrainingArguments(output_dir=str(tmp_path), per_device_train_batch_size=4, learning_rate=5e-3, warmup_steps=1,
                                    max_steps=12, logging_steps=1, save_steps=12, predict_with_generate=False)
    tr = Seq2SeqTrainer(args=args, model=model, train_dataset=ds, eval_dataset=ds,
                        data_collator=DataCollatorSpeechSeq2SeqWithPadding(processor=None, decoder_start_token_id=50258), tokenizer=None)
    tr.train()
    losses = [h["loss"] for h in tr.log_history]
    assert losses[-1] < losses[0] - 0.05, losses
    ck = torch.load(tmp_path / "lora_adapters.pt")
    assert set(ck["lora"]) == {k for k in model.encoder.state_dict() if "lora_" in k} and ck["r"] == 8


def test_sharded_gradients_average_to_full_batch_gradient():
    cfg = wts.config("mini", True)
    model = _model(cfg)
    for p in model.lora_parameters():      # non-zero B so both adapter matrices get gradient
        if p.shape[1] == 8:
            p.data.copy_(torch.from_numpy((0.02 * wts.unit_variates("b", p.numel(), 1)).reshape(p.shape).astype(np.float32)))
    full = _batch(cfg, 4)
    full["labels"][0, 4:] = 7           # equal token counts per shard so that mean-of-means == global mean

    def grads(batch):
        model.zero_grad()
        model(input_features=batch["input_features"].cuda(), labels=batch["labels"].cuda()).loss.backward()
        return torch.cat([p.grad.flatten() for p in model.lora_parameters()]).clone()

    g_full = grads(full)
    halves = [{k: v[i:i + 2] for k, v in full.items()} for i in (0, 2)]
    g_avg = (grads(halves[0]) + grads(halves[1])) / 2
    rel = (g_full - g_avg).abs().max() / g_full.abs().max()
    assert rel < 1e-3, float(rel)


def test_native_cross_kv_projection_matches_torch_decoder():
    """The fused native cross-attention K/V projection (one GEMM forward, one backward) gives the loss and the adapter
    gradients of the all-torch decoder."""
    from mlx8_ws_audio_transformer_amd.finetune import WhisperLoRAModel
    cfg = wts.config("mini", True)
    b = _batch(cfg, 3)
    out = {}
    for native in (True, False):
        model = WhisperLoRAModel(cfg, wts.LoraSpec(r=8, alpha=16.0), decoder_layers=2, native_cross_kv=native)
        for p in model.lora_parameters():
            if p.shape[1] == 8:
                with torch.no_grad():
                    p.copy_(torch.from_numpy(0.05 * wts.unit_variates("kvtest", p.numel(), 1).reshape(p.shape).astype(np.float32)))
        res = model(input_features=b["input_features"].cuda(), labels=b["labels"].cuda())
        res.loss.backward()
        out[native] = (float(res.loss.detach()), torch.cat([p.grad.flatten() for p in model.lora_parameters()]).cpu())
    assert abs(out[True][0] - out[False][0]) < 2e-4 * abs(out[False][0])
    ref = out[False][1]
    assert float((out[True][1] - ref).abs().max()) < 2e-3 * float(ref.abs().max())
