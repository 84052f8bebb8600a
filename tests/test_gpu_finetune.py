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
    losses = [h["loss"] for h in tr.log_history if "loss" in h]
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
        model = WhisperLoRAModel(cfg, wts.LoraSpec(r=8, alpha=16.0), decoder_layers=2, native_cross_kv=native, native_decoder=False)
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


def _golden_model():
    """WhisperLoRAModel with the deterministic weights of tests/golden/decoder.npz (tools/make_golden.py gen_decoder)."""
    from mlx8_ws_audio_transformer_amd.finetune import WhisperLoRAModel
    cfg = wts.config("mini", True)
    model = WhisperLoRAModel(cfg, wts.LoraSpec(r=8, alpha=16.0), decoder_layers=2, vocab=512, max_target_positions=64)
    model.config.decoder_start_token_id, model.config.pad_token_id, model.config.eos_token_id = 1, 0, 2
    We = wts.init_encoder_weights(cfg, seed=0, profile="test")
    Wd = wts.init_decoder_weights(cfg.d_model, 2, cfg.ffn, 512, 64, seed=0)
    model.encoder.load_state_dict({k: torch.from_numpy(v) for k, v in We.items()}, strict=False)      # LoRA B = 0: adapters are inert
    model.decoder.load_state_dict({k: torch.from_numpy(v) for k, v in Wd.items()}, strict=True)
    mel = oracle_mel.whisper_logmel(piano_clips_f32(2), n_samples=2 * cfg.max_source_positions * 160)
    return model.eval(), torch.from_numpy(mel).cuda()


def test_step_loss_logits_and_greedy_tokens_match_reference():
    """fineTune.py's forward (a9: shift labels, encoder, decoder, tied projection, CE) and greedy decoding against
    WhisperForConditionalGeneration on the same weights."""
    from tests.util import golden
    G = golden("decoder.npz")
    model, mel = _golden_model()
    with torch.no_grad():
        out = model(input_features=mel, labels=torch.from_numpy(G["labels"]).cuda())
    np.testing.assert_allclose(out.encoder_last_hidden_state[:, :4].cpu().numpy(), G["encoder_head"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(out.logits.float().cpu().numpy(), G["logits"], rtol=0, atol=2e-3)
    assert abs(float(out.loss) - float(G["loss"])) < 1e-3
    ids = model.generate(mel, max_length=G["greedy_ids"].shape[1]).cpu().numpy()
    np.testing.assert_array_equal(ids, G["greedy_ids"])


def test_evaluate_with_generate_and_wer():
    from mlx8_ws_audio_transformer_amd.collator import DataCollatorSpeechSeq2SeqWithPadding
    from mlx8_ws_audio_transformer_amd.finetune import Seq2SeqTrainer, Seq2SeqTrainingArguments, wer
    from tests.util import golden
    G = golden("decoder.npz")
    model, mel = _golden_model()
    greedy = G["greedy_ids"]
    # labels = what the model itself decodes for clip 0, and a deliberately wrong sequence for clip 1
    ds = [{"input_features": mel[0].cpu().numpy(), "labels": greedy[0, :8].tolist()},
          {"input_features": mel[1].cpu().numpy(), "labels": [1, 5, 6, 7, 8, 9, 10, 11]}]

    def decode(rows):
        return [" ".join(str(t) for t in r if t not in (-100, 0, 1, 2)) for r in rows]

    def compute_metrics(pred):                      # the body of fineTune.py:145-158 with a toy tokenizer
        lab = np.where(pred.label_ids == -100, 0, pred.label_ids)
        return {"wer": 100 * wer(decode(lab), decode(pred.predictions))}

    args = Seq2SeqTrainingArguments(per_device_eval_batch_size=2, generation_max_length=8, predict_with_generate=True)
    tr = Seq2SeqTrainer(args=args, model=model, eval_dataset=ds, compute_metrics=compute_metrics,
                        data_collator=DataCollatorSpeechSeq2SeqWithPadding(processor=None, decoder_start_token_id=1))
    m = tr.evaluate()
    assert m["eval_loss"] > 0
    # clip 0 is decoded exactly (0 errors of 7 words), clip 1 shares no word with its reference (7 errors of 7)
    assert abs(m["eval_wer"] - 50.0) < 1e-6, m


def test_gradient_accumulation_equals_one_big_batch():
    """gradient_accumulation_steps = 2 over two half batches takes the same optimizer step as one full batch
    (mean-of-means = mean here: equal micro-batch sizes and the same number of label tokens per clip)."""
    from mlx8_ws_audio_transformer_amd.finetune import Seq2SeqTrainer, Seq2SeqTrainingArguments
    cfg = wts.config("mini", True)
    b = _batch(cfg, 4)
    b["labels"] = b["labels"].clone()
    b["labels"][b["labels"] == -100] = 7                      # same token count in every row: the two loss means weigh equally
    steps = {}
    for ga in (1, 2):
        model = _model(cfg)
        for p in model.lora_parameters():
            if p.shape[1] == 8:
                with torch.no_grad():
                    p.copy_(torch.from_numpy(0.05 * wts.unit_variates("ga", p.numel(), 1).reshape(p.shape).astype(np.float32)))
        args = Seq2SeqTrainingArguments(per_device_train_batch_size=4 // ga, gradient_accumulation_steps=ga, learning_rate=1e-2, warmup_steps=0,
                                        max_steps=1, predict_with_generate=False, max_grad_norm=0.0)
        tr = Seq2SeqTrainer(args=args, model=model)
        before = torch.cat([p.detach().flatten().clone() for p in model.lora_parameters()])
        if ga == 1:
            tr.training_step(b)
        else:
            tr.training_step([{k: v[:2] for k, v in b.items()}, {k: v[2:] for k, v in b.items()}])
        steps[ga] = torch.cat([p.detach().flatten() for p in model.lora_parameters()]) - before
    assert float(steps[1].abs().max()) > 0
    assert float((steps[1] - steps[2]).abs().max()) < 2e-3 * float(steps[1].abs().max())


@pytest.mark.parametrize("fmt", ["safetensors", "bin"])
def test_from_pretrained_directory_reproduces_the_reference_vectors(tmp_path, fmt):
    """VERDICT r2 missing #2: the reference's inference scripts load the fine-tuned checkpoint BY PATH (AB/wavToWhisper.py:39,47,
    AB/fineTuneMidiTester.py:20-21).  A directory written in the layout `save_pretrained` / `trainer.save_model()` produce (config.json +
    model.safetensors, or pytorch_model.bin as transformers 4.35 writes) loads through `WhisperLoRAModel.from_pretrained(path)` -- no
    `transformers` involved -- and gives WhisperForConditionalGeneration's loss, logits and greedy ids on those weights (decoder.npz)."""
    from mlx8_ws_audio_transformer_amd import checkpoint as ck
    from mlx8_ws_audio_transformer_amd.finetune import WhisperLoRAModel
    from tests.util import golden
    G = golden("decoder.npz")
    cfg = wts.config("mini", True)
    We = {k: torch.from_numpy(v) for k, v in wts.init_encoder_weights(cfg, 0, "test").items()}
    Wd = {k: torch.from_numpy(v) for k, v in wts.init_decoder_weights(cfg.d_model, 2, cfg.ffn, 512, 64, 0).items()}
    hf = {"architectures": ["WhisperForConditionalGeneration"], "model_type": "whisper", "d_model": cfg.d_model, "encoder_layers": cfg.layers,
          "encoder_attention_heads": cfg.heads, "encoder_ffn_dim": cfg.ffn, "decoder_layers": 2, "decoder_attention_heads": cfg.heads, "decoder_ffn_dim": cfg.ffn,
          "num_mel_bins": 80, "max_source_positions": cfg.max_source_positions, "max_target_positions": 64, "vocab_size": 512,
          "decoder_start_token_id": 1, "pad_token_id": 0, "eos_token_id": 2}
    path = ck.save_pretrained_dir(str(tmp_path / "whisper-small-hi"), hf, We, Wd, fmt=fmt)
    model = WhisperLoRAModel.from_pretrained(path).eval()                  # inference: no adapters, precision chosen from the checkpoint
    assert model.encoder.lora is None and not model.encoder.trainable and model.config.decoder_start_token_id == 1 and model.config.eos_token_id == 2
    mel = torch.from_numpy(oracle_mel.whisper_logmel(piano_clips_f32(2), n_samples=2 * cfg.max_source_positions * 160)).cuda()
    with torch.no_grad():
        out = model(input_features=mel, labels=torch.from_numpy(G["labels"]).cuda())
    assert model.encoder.precision in ("f16f8", "fp16x3")
    np.testing.assert_allclose(out.encoder_last_hidden_state[:, :4].cpu().numpy(), G["encoder_head"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(out.logits.float().cpu().numpy(), G["logits"], rtol=0, atol=2e-3)
    assert abs(float(out.loss) - float(G["loss"])) < 1e-3
    np.testing.assert_array_equal(model.generate(mel, max_length=G["greedy_ids"].shape[1]).cpu().numpy(), G["greedy_ids"])
    # with adapters on top of the loaded base (B = 0 at init: inert), and the merged directory written back by save_pretrained
    tuned = WhisperLoRAModel.from_pretrained(path, lora=wts.LoraSpec(r=8, alpha=16.0))
    assert tuned.encoder.trainable and tuned.encoder.precision == "bf16x3"
    with torch.no_grad():
        for p in tuned.lora_parameters():
            if p.shape[1] == 8:            # lora_B [out, r]
                p.copy_(torch.from_numpy(0.05 * wts.unit_variates("ckpt", p.numel(), 2).reshape(p.shape).astype(np.float32)))
        loss_tuned = float(tuned(input_features=mel, labels=torch.from_numpy(G["labels"]).cuda()).loss)
    merged = WhisperLoRAModel.from_pretrained(tuned.save_pretrained(str(tmp_path / "merged")), precision="bf16x3").eval()
    with torch.no_grad():
        loss_merged = float(merged(input_features=mel, labels=torch.from_numpy(G["labels"]).cuda()).loss)
    assert abs(loss_tuned - float(G["loss"])) > 1e-4 and abs(loss_merged - loss_tuned) < 2e-3 * abs(loss_tuned)


def test_transcribe_from_a_checkpoint_directory(tmp_path):
    """a14 on the artefact the reference produces: `transcribe_audio_FT` with a model loaded from a directory on disk (wavToWhisper.py:44-70)."""
    import wave
    from mlx8_ws_audio_transformer_amd import checkpoint as ck, synth
    from mlx8_ws_audio_transformer_amd.feature_extraction import WhisperProcessor
    from mlx8_ws_audio_transformer_amd.finetune import WhisperLoRAModel
    from mlx8_ws_audio_transformer_amd.transcribe import NoteTokenizer, transcribe_audio_FT
    tok = NoteTokenizer()
    cfg = wts.config("mini", False)              # the processor pads to 30 s: 3000 frames -> 1500 positions
    We = {k: torch.from_numpy(v) for k, v in wts.init_encoder_weights(cfg, 0, "test").items()}
    Wd = {k: torch.from_numpy(v) for k, v in wts.init_decoder_weights(cfg.d_model, 1, cfg.ffn, tok.vocab_size, 64, 0).items()}
    hf = {"d_model": cfg.d_model, "encoder_layers": cfg.layers, "encoder_attention_heads": cfg.heads, "encoder_ffn_dim": cfg.ffn, "decoder_layers": 1,
          "decoder_attention_heads": cfg.heads, "decoder_ffn_dim": cfg.ffn, "num_mel_bins": 80, "max_source_positions": cfg.max_source_positions,
          "max_target_positions": 64, "vocab_size": tok.vocab_size, "decoder_start_token_id": tok.bos_token_id, "pad_token_id": tok.pad_token_id,
          "eos_token_id": tok.eos_token_id}
    path = ck.save_pretrained_dir(str(tmp_path / "whisper-small-piano"), hf, We, Wd)
    pcm = synth.synth_clips_i16(1, seed=1234, first=2)[0]
    wav = tmp_path / "clip.wav"
    with wave.open(str(wav), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(pcm.tobytes())
    model = WhisperLoRAModel.from_pretrained(path).eval()
    results = []
    text = transcribe_audio_FT(wav, results, model, WhisperProcessor(tokenizer=tok), max_length=10)
    assert isinstance(text, str) and results[0]["Transcription"] == text and (tmp_path / "clip.text").exists()
    again = WhisperLoRAModel.from_pretrained(path, precision="fp16x3").eval()         # deterministic: a second load decodes the same tokens
    r2 = []
    assert transcribe_audio_FT(wav, r2, again, WhisperProcessor(tokenizer=tok), write_text=False, max_length=10) == text
