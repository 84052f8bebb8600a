"""GPU parity: HIP log-mel kernels (through the C-ABI) vs the oracle and the committed reference vectors."""
import numpy as np
import pytest
import torch

from oracle import logmel as oracle
from tests.util import golden, logmel_inputs, piano_clips_f32

pytestmark = pytest.mark.gpu

MEL_TOL = 1e-5  # north-star tolerance for mel bins (fp32)


@pytest.fixture(scope="module")
def fe():
    from mlx8_ws_audio_transformer_amd.feature_extraction import WhisperFeatureExtractor
    return WhisperFeatureExtractor()


@pytest.mark.parametrize("name", ["noise", "tone", "zeros", "short", "piano"])
def test_whisper_logmel_matches_golden_and_oracle(fe, name):
    G = golden("logmel_whisper.npz")
    clip = logmel_inputs()[name]
    out = fe(clip, sampling_rate=16000, return_tensors="np")["input_features"]
    assert out.shape == (1, 80, 3000) and out.dtype == np.float32
    np.testing.assert_allclose(out[0, :, :404], G[f"{name}_np_live"], rtol=0, atol=MEL_TOL)
    np.testing.assert_allclose(out[0], oracle.whisper_logmel([clip])[0], rtol=0, atol=MEL_TOL)


def test_batch_ragged_lengths_and_per_clip_max(fe):
    ins = logmel_inputs()
    clips = [ins["tone"], ins["short"] * 0.01, ins["piano"][:12345], np.zeros(7, np.float32), ins["noise"]]
    out = fe(clips, sampling_rate=16000, return_tensors="pt")["input_features"]
    assert out.device.type == "cpu" and tuple(out.shape) == (5, 80, 3000)
    np.testing.assert_allclose(out.numpy(), oracle.whisper_logmel(clips), rtol=0, atol=MEL_TOL)


def test_int16_pcm_path_and_device_entry_point():
    from mlx8_ws_audio_transformer_amd import synth
    from mlx8_ws_audio_transformer_amd.feature_extraction import logmel_whisper_device
    pcm = synth.synth_clips_i16(6, seed=1234, first=10)
    dev = torch.from_numpy(pcm).cuda()
    out = logmel_whisper_device(dev).cpu().numpy()
    ref = oracle.whisper_logmel([synth.pcm_i16_to_f32(c) for c in pcm])
    np.testing.assert_allclose(out, ref, rtol=0, atol=MEL_TOL)


def test_keep_on_device_returns_the_same_features_in_hbm(fe):
    clips = piano_clips_f32(2)
    host = fe(clips, sampling_rate=16000, return_tensors="pt")["input_features"]
    dev = fe(clips, sampling_rate=16000, return_tensors="pt", keep_on_device=True)["input_features"]
    assert dev.is_cuda and not host.is_cuda and torch.equal(dev.cpu(), host)
    with pytest.raises(ValueError):
        fe(clips, sampling_rate=16000, return_tensors="np", keep_on_device=True)


def test_full_30s_clip_and_truncation(fe):
    x = np.concatenate([logmel_inputs()["tone"]] * 8)[:500000]   # longer than 30 s: truncated like the reference
    out = fe(x, sampling_rate=16000, return_tensors="np")["input_features"][0]
    np.testing.assert_allclose(out, oracle.whisper_logmel([x])[0], rtol=0, atol=MEL_TOL)


def test_trimmed_mode(fe):
    G = golden("logmel_whisper.npz")
    out = fe(logmel_inputs()["tone"], sampling_rate=16000, return_tensors="np", max_length=64000)["input_features"][0]
    assert out.shape == (80, 400)
    np.testing.assert_allclose(out, G["tone_torch_trimmed"], rtol=0, atol=MEL_TOL)
    np.testing.assert_allclose(out, oracle.whisper_logmel([logmel_inputs()["tone"]], n_samples=64000)[0], rtol=0, atol=MEL_TOL)


def test_reference_error_behaviour(fe):
    with pytest.raises(ValueError, match="16000"):
        fe(np.zeros(100, np.float32), sampling_rate=8000)
    with pytest.raises(ValueError, match="mono"):
        fe(np.zeros((2, 2, 100), np.float32), sampling_rate=16000)
    m = fe(logmel_inputs()["short"], sampling_rate=16000, return_tensors="np", return_attention_mask=True)["attention_mask"]
    assert m.shape == (1, 3000) and m[0, :100].all() and m.sum() == 100


@pytest.mark.parametrize("n_mels,hop", [(80, 512), (128, 512), (128, 128), (64, 512)])
def test_urbansound_logmel(n_mels, hop):
    from mlx8_ws_audio_transformer_amd import urbansound
    G3 = golden("logmel_urbansound.npz")
    tone = logmel_inputs()["tone"]
    out = urbansound.mel_spectrogram_log(torch.from_numpy(tone), n_fft=1024, hop_length=hop, n_mels=n_mels).cpu().numpy()
    ref = oracle.urbansound_logmel(tone, n_fft=1024, hop=hop, n_mels=n_mels)
    assert out.shape == ref.shape
    np.testing.assert_allclose(out, ref, rtol=0, atol=MEL_TOL)
    np.testing.assert_allclose(out, G3[f"mels{n_mels}_hop{hop}"], rtol=0, atol=2e-4)  # fp32 torch.stft restatement


def test_urbansound_batch_and_prepare():
    from mlx8_ws_audio_transformer_amd import urbansound
    clips = piano_clips_f32(3)
    st = torch.from_numpy(np.stack([clips[0][:30000], clips[1][:30000]]))       # "stereo", short
    w = urbansound.prepare_waveform(st)
    assert tuple(w.shape) == (1, 64000)
    np.testing.assert_array_equal(w[0].numpy(), oracle.urbansound_prepare(st.numpy()))
    batch = torch.from_numpy(np.stack(clips))
    out = urbansound.mel_spectrogram_log(batch, n_mels=80).cpu().numpy()
    for i in range(3):
        np.testing.assert_allclose(out[i], oracle.urbansound_logmel(clips[i], n_mels=80), rtol=0, atol=MEL_TOL)


def test_prepare_dataset_through_the_native_processor():
    """Scope row a1 end to end: the reference's `prepare_dataset` (AB/fineTune.py:85-92) bound to the native `WhisperProcessor` (HIP log-mel +
    the note tokenizer) on a seeded clip -> `input_features` [80, 3000] within 1e-5 of the oracle, `labels` exactly the tokenizer's ids; then
    through the reference's collator."""
    from mlx8_ws_audio_transformer_amd.collator import DataCollatorSpeechSeq2SeqWithPadding, make_prepare_dataset
    from mlx8_ws_audio_transformer_amd.feature_extraction import WhisperProcessor
    from mlx8_ws_audio_transformer_amd.transcribe import NoteTokenizer
    from tests.util import piano_clips_f32
    tok = NoteTokenizer()
    prep = make_prepare_dataset(WhisperProcessor(tokenizer=tok))
    clips = piano_clips_f32(2, 5)
    texts = ["<|MIDI|> G#6 F2 C4 <|/MIDI|>", "<|MIDI|> A0 <|/MIDI|>"]
    rows = [prep({"audio": {"array": c, "sampling_rate": 16000}, "sentence": t}) for c, t in zip(clips, texts)]
    ref = oracle.whisper_logmel(clips)
    for r, m, t in zip(rows, ref, texts):
        f = np.asarray(r["input_features"])
        assert f.shape == (80, 3000) and f.dtype == np.float32
        assert np.abs(f - m).max() <= 1e-5
        assert r["labels"] == tok(t)["input_ids"] and r["labels"][0] == tok.bos_token_id
    batch = DataCollatorSpeechSeq2SeqWithPadding(processor=WhisperProcessor(tokenizer=tok), decoder_start_token_id=tok.bos_token_id)(rows)
    assert tuple(batch["input_features"].shape) == (2, 80, 3000)
    assert batch["labels"].tolist() == [tok(texts[0])["input_ids"][1:], tok(texts[1])["input_ids"][1:] + [-100, -100]]     # BOS stripped, -100 padding
    with pytest.raises(ValueError):
        prep({"audio": {"array": clips[0], "sampling_rate": 44100}, "sentence": texts[0]})


def test_real_recording_end_to_end(fe):
    """VERDICT r2 missing #3: the reference's own sample recording.  Interleaved stereo int16 -> `awt_prepare_waveform` (channel mean) ->
    HIP log-mel within 1e-5 of the reference's float64 path; then Whisper-tiny on the HIP features within 1e-3 of `WhisperEncoder`."""
    from mlx8_ws_audio_transformer_amd import urbansound, weights as wts
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    from tests.util import real_audio
    pcm, mono, R = real_audio()
    wav = urbansound.prepare_waveform(torch.from_numpy(pcm), sample_rate=16000, target_rate=16000, interleaved=True, n_out=64000)[0].cpu().numpy()
    assert np.abs(wav - mono).max() <= 1e-7
    feats = fe(wav, sampling_rate=16000, return_tensors="np")["input_features"]
    assert feats.shape == (1, 80, 3000)
    assert np.abs(feats[0][:, :404] - R["np_live"]).max() <= MEL_TOL
    assert np.abs(feats[0][:, 404:] - float(R["np_padconst"])).max() <= MEL_TOL
    cfg = wts.config("tiny", False)
    for precision in (None, "fp16x3", "bf16x3"):
        enc = NativeWhisperEncoder(cfg, precision=precision, seed=0, init_profile="test").eval()
        out = enc(torch.from_numpy(feats).cuda()).last_hidden_state.cpu().numpy()
        for key, sl in (("last_head", slice(0, 4)), ("last_live", slice(196, 204)), ("last_tail", slice(-4, None))):
            assert np.abs(out[:, sl] - R["tiny/" + key]).max() <= 1e-3, (precision, key)
