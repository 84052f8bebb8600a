"""CPU tests of the host-side mirrors that need no GPU: UrbanSound dataset schema, trainer-surface helpers, synthesiser."""
import numpy as np
import pytest
import torch

from mlx8_ws_audio_transformer_amd import synth
from mlx8_ws_audio_transformer_amd.finetune import Seq2SeqTrainingArguments, linear_schedule, shift_tokens_right


def test_urbansound_dataset_reads_reference_parquet_schema(tmp_path):
    import pandas as pd
    from mlx8_ws_audio_transformer_amd import urbansound
    rows = []
    for i in range(6):
        mel = torch.arange(80 * 126, dtype=torch.float32).reshape(80, 126) + i
        rows.append(urbansound.record_for_parquet(f"audio/fold{1 + i % 3}/f{i}.wav", 1 + i % 3, i % 10, f"class{i % 10}", mel))
    path = tmp_path / urbansound.get_processed_parquet_filename(80, 512)
    pd.DataFrame(rows).to_parquet(path, index=False)          # the reference writes with DataFrame.to_parquet too (spectrogram.py:177-181)
    assert set(pd.read_parquet(path).columns) == {"rel_path", "fold", "class_id", "class_name", "log_mel_flat", "log_mel_shape"}
    ds = urbansound.UrbanSoundDataSet(parquet_path=str(path), folds=[1, 3])
    assert len(ds) == 4 and hasattr(ds, "df") and ds.n_mels == urbansound.N_MELS
    x, y = ds[1]
    assert isinstance(x, torch.Tensor) and x.dtype == torch.float32 and tuple(x.shape) == (80, 126) and isinstance(y, int)
    assert float(x[0, 0]) == float(ds.df.iloc[1]["log_mel_flat"][0])


def test_prepare_waveform_matches_reference_rules():
    from mlx8_ws_audio_transformer_amd import urbansound
    st = torch.stack([torch.ones(1000), 3 * torch.ones(1000)])
    w = urbansound.prepare_waveform(st)
    assert tuple(w.shape) == (1, 64000) and torch.all(w[0, :1000] == 2) and torch.all(w[0, 1000:] == 0)
    assert tuple(urbansound.prepare_waveform(torch.ones(70000)).shape) == (1, 64000)


def test_shift_tokens_right_and_schedule():
    labels = torch.tensor([[5, 6, -100], [7, -100, -100]])
    out = shift_tokens_right(labels, 50257, 50258)
    assert out.tolist() == [[50258, 5, 6], [50258, 7, 50257]]
    assert linear_schedule(0, 1, 50) == 0.0 and linear_schedule(1, 1, 50) == 1.0 and abs(linear_schedule(25, 1, 50) - 25 / 49) < 1e-12
    assert linear_schedule(50, 1, 50) == 0.0


def test_training_arguments_keep_reference_field_names():
    # every keyword /root/reference/AB/fineTune.py:162-183 passes must be accepted
    a = Seq2SeqTrainingArguments(output_dir="./whisper-small-hi", per_device_train_batch_size=16, gradient_accumulation_steps=1,
                                 learning_rate=1e-5, warmup_steps=1, max_steps=50, gradient_checkpointing=True, fp16=False,
                                 evaluation_strategy="steps", per_device_eval_batch_size=8, predict_with_generate=True,
                                 generation_max_length=225, save_steps=50, eval_steps=10, logging_steps=10, report_to=["wandb"],
                                 load_best_model_at_end=True, metric_for_best_model="wer", greater_is_better=False, push_to_hub=False)
    assert a.max_steps == 50 and a.learning_rate == 1e-5


def test_synth_follows_reference_distributions():
    for i in range(50):
        notes = synth.clip_notes(1234, i)
        assert len(notes) == 5
        t_prev = 0.0
        for start, dur, pitch in notes:
            assert 21 <= pitch <= 108 and dur in synth.DURATIONS
            assert round(start - t_prev, 6) >= 0.1 - 1e-9       # a gap precedes every note
            t_prev = start + dur
    assert synth.clip_label(1234, 0).startswith("<|MIDI|> ") and synth.clip_label(1234, 0).endswith(" <|/MIDI|>")
    a, b = synth.synth_clips_i16(2, seed=1234), synth.synth_clips_i16(1, seed=1234, first=1)
    assert a.dtype == np.int16 and a.shape == (2, 64000) and np.array_equal(a[1], b[0])
    assert synth.note_number_to_name(69) == "A4" and synth.note_number_to_name(21) == "A0"


def test_classifier_keeps_reference_parameter_names_and_has_no_cpu_fallback():
    import pytest
    from mlx8_ws_audio_transformer_amd.urbansound_classifier import TransformerUrbanSound8KClassifier
    from oracle.urbansound_classifier import ReferenceTransformerClassifier
    nat, ref = TransformerUrbanSound8KClassifier(n_mels=80), ReferenceTransformerClassifier(n_mels=80)
    assert list(nat.state_dict()) == list(ref.state_dict())            # a reference checkpoint loads with load_state_dict
    assert {k: tuple(v.shape) for k, v in nat.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    for mode in (nat.train(), nat.eval()):                              # both paths are libawt calls: CPU tensors are refused, not routed to torch
        with pytest.raises(ValueError, match="cuda device"):
            mode(torch.zeros(1, 80, 126))
    with pytest.raises(ValueError):
        TransformerUrbanSound8KClassifier(dim=128, heads=1)            # head_dim 128 > the kernel's 64


# ---------------------------------------------------------------- a14: tokenizer stand-in of the transcribe / tester loop (host side)
def test_note_tokenizer_roundtrip_and_collator_padding():
    from mlx8_ws_audio_transformer_amd import synth
    from mlx8_ws_audio_transformer_amd.collator import DataCollatorSpeechSeq2SeqWithPadding
    from mlx8_ws_audio_transformer_amd.feature_extraction import WhisperProcessor
    from mlx8_ws_audio_transformer_amd.transcribe import NoteTokenizer
    tok = NoteTokenizer()
    label = synth.clip_label(1234, 0)                       # "<|MIDI|> G#6 F2 ... <|/MIDI|>" like AB/synthDataset.py:82
    ids = tok(label)["input_ids"]
    assert ids[0] == tok.bos_token_id and ids[-1] == tok.eos_token_id and tok.unk_token_id not in ids
    assert tok.decode(ids, skip_special_tokens=True) == label
    assert tok.batch_decode([ids, ids[:4]], skip_special_tokens=True)[1] == " ".join(label.split()[:3])
    both = tok([label, "<|MIDI|> C4 <|/MIDI|>"])["input_ids"]
    assert len(both) == 2 and len(both[1]) == 5
    proc = WhisperProcessor(tokenizer=tok)
    assert proc(text=label)["labels"] == ids               # the processor's text branch (fineTune.py:88)
    coll = DataCollatorSpeechSeq2SeqWithPadding(processor=proc, decoder_start_token_id=tok.bos_token_id)
    feats = [{"input_features": np.zeros((80, 8), np.float32), "labels": both[0]}, {"input_features": np.zeros((80, 8), np.float32), "labels": both[1]}]
    batch = coll(feats)
    assert batch["labels"].shape == (2, len(both[0]) - 1)    # every row starts with BOS: the collator strips it (fineTune.py:114-115)
    assert (batch["labels"][1, len(both[1]) - 1:] == -100).all()


def test_automatic_precision_choice_reads_the_weights_only():
    """NativeWhisperEncoder(precision=None).choose_precision(): pure function of the parameters (runs on CPU tensors too)."""
    from mlx8_ws_audio_transformer_amd import weights as wts
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    cfg = wts.config("mini", True)
    enc = NativeWhisperEncoder(cfg, device="cpu", seed=0, init_profile="hf")
    assert enc.choose_precision() == "f16f8" and enc.precision_report["row_norm_ratio"] < 5
    W = wts.with_outlier_channels(wts.init_encoder_weights(cfg, 0, "test"), cfg, seed=0)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    assert enc.choose_precision() == "fp16x3"
    rep = enc.precision_report
    assert rep["layernorm_gain_ratio"] > 8 and rep["precision"] == "fp16x3"
