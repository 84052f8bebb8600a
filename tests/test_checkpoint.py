"""CPU: the checkpoint-directory reader behind `WhisperLoRAModel.from_pretrained(path)` -- the artefact the reference's fine-tune leaves behind
(AB/fineTune.py:200 `trainer.save_model()`) and its inference scripts load by path (AB/wavToWhisper.py:39,47; AB/fineTuneMidiTester.py:20-21)."""
import json
import os
import struct

import numpy as np
import pytest
import torch

from mlx8_ws_audio_transformer_amd import checkpoint as ck, weights as wts

HF_CFG = {"architectures": ["WhisperForConditionalGeneration"], "model_type": "whisper", "d_model": 128, "encoder_layers": 2, "encoder_attention_heads": 2,
          "encoder_ffn_dim": 512, "decoder_layers": 2, "decoder_attention_heads": 2, "decoder_ffn_dim": 512, "num_mel_bins": 80, "max_source_positions": 200,
          "max_target_positions": 64, "vocab_size": 512, "decoder_start_token_id": 1, "pad_token_id": 0, "eos_token_id": 2}


def _state():
    cfg = wts.config("mini", True)
    We = {k: torch.from_numpy(v) for k, v in wts.init_encoder_weights(cfg, 0, "test").items()}
    Wd = {k: torch.from_numpy(v) for k, v in wts.init_decoder_weights(cfg.d_model, 2, cfg.ffn, 512, 64, 0).items()}
    return cfg, We, Wd


def test_safetensors_reader_matches_the_published_format(tmp_path):
    t = {"a": torch.randn(3, 5), "b.c": torch.randn(7).half(), "i": torch.arange(4), "bf": torch.randn(2, 2).bfloat16(), "empty": torch.empty(0, 3)}
    p = str(tmp_path / "x.safetensors")
    ck.write_safetensors(p, t, {"format": "pt"})
    raw = open(p, "rb").read()
    (n,) = struct.unpack("<Q", raw[:8])
    header = json.loads(raw[8: 8 + n])
    assert header["__metadata__"] == {"format": "pt"} and header["a"] == {"dtype": "F32", "shape": [3, 5], "data_offsets": header["a"]["data_offsets"]}
    r = ck.read_safetensors(p)
    assert set(r) == set(t) and all(torch.equal(r[k], t[k]) and r[k].dtype == t[k].dtype for k in t)
    assert set(ck.read_safetensors(p, keys=["a"])) == {"a"}
    st = pytest.importorskip("safetensors.torch")              # cross-check against the format's own implementation where it is installed
    assert all(torch.equal(v, t[k]) for k, v in st.load_file(p).items())
    st.save_file({k: v.contiguous() for k, v in t.items()}, str(tmp_path / "y.safetensors"))
    r2 = ck.read_safetensors(str(tmp_path / "y.safetensors"))
    assert all(torch.equal(r2[k], t[k]) for k in t)
    open(str(tmp_path / "bad.safetensors"), "wb").write(struct.pack("<Q", 1 << 40) + b"{}")
    with pytest.raises(ValueError, match="header length"):
        ck.read_safetensors(str(tmp_path / "bad.safetensors"))


@pytest.mark.parametrize("fmt", ["safetensors", "bin"])
def test_checkpoint_directory_roundtrip(tmp_path, fmt):
    cfg, We, Wd = _state()
    d = ck.save_pretrained_dir(str(tmp_path / "whisper-small-hi"), HF_CFG, We, Wd, fmt=fmt)
    assert os.path.exists(os.path.join(d, "config.json")) and os.path.exists(os.path.join(d, "model.safetensors" if fmt == "safetensors" else "pytorch_model.bin"))
    hf, enc, dec = ck.load_checkpoint_dir(d)
    assert ck.encoder_config_from_hf(hf).d_model == cfg.d_model and ck.encoder_config_from_hf(hf).max_source_positions == 200
    assert set(enc) == set(We) and set(dec) == set(Wd)
    assert all(torch.equal(enc[k], We[k]) for k in We) and all(torch.equal(dec[k], Wd[k]) for k in Wd)


def test_half_precision_and_sharded_checkpoints(tmp_path):
    cfg, We, Wd = _state()
    d = ck.save_pretrained_dir(str(tmp_path / "fp16"), HF_CFG, We, Wd, dtype=torch.float16)
    _, enc, _ = ck.load_checkpoint_dir(d)
    k = "layers.0.fc1.weight"
    assert enc[k].dtype == torch.float32 and torch.equal(enc[k], We[k].half().float())       # fp32 tensors holding half-precision values
    # sharded layout: model.safetensors.index.json + two shards
    sh = tmp_path / "sharded"
    sh.mkdir()
    json.dump(HF_CFG, open(sh / "config.json", "w"))
    full = {"model.encoder." + k: v for k, v in We.items()}
    full.update({"model.decoder." + k: v for k, v in Wd.items()})
    names = sorted(full)
    parts = {"model-00001-of-00002.safetensors": names[: len(names) // 2], "model-00002-of-00002.safetensors": names[len(names) // 2:]}
    for f, ks in parts.items():
        ck.write_safetensors(str(sh / f), {k: full[k] for k in ks})
    json.dump({"metadata": {}, "weight_map": {k: f for f, ks in parts.items() for k in ks}}, open(sh / "model.safetensors.index.json", "w"))
    _, enc2, dec2 = ck.load_checkpoint_dir(str(sh))
    assert all(torch.equal(enc2[k], We[k]) for k in We) and all(torch.equal(dec2[k], Wd[k]) for k in Wd)


def test_errors_are_loud(tmp_path):
    cfg, We, Wd = _state()
    with pytest.raises(FileNotFoundError):
        ck.load_checkpoint_dir(str(tmp_path / "nope"))
    e = tmp_path / "empty"
    e.mkdir()
    with pytest.raises(FileNotFoundError, match="config.json"):
        ck.load_checkpoint_dir(str(e))
    json.dump(HF_CFG, open(e / "config.json", "w"))
    with pytest.raises(FileNotFoundError, match="model.safetensors"):
        ck.load_checkpoint_dir(str(e))
    ck.write_safetensors(str(e / "model.safetensors"), {"lm_head.weight": torch.zeros(2, 2)})
    with pytest.raises(KeyError, match="unexpected tensor"):
        ck.load_checkpoint_dir(str(e))
    # an untied projection is not a Whisper checkpoint
    u = tmp_path / "untied"
    ck.save_pretrained_dir(str(u), HF_CFG, We, Wd, fmt="bin")
    sd = torch.load(u / "pytorch_model.bin", weights_only=True)
    sd["proj_out.weight"] = sd["proj_out.weight"] + 1
    torch.save(sd, u / "pytorch_model.bin")
    with pytest.raises(ValueError, match="ties"):
        ck.load_checkpoint_dir(str(u))
