"""Local Whisper checkpoint directories: `config.json` + `model.safetensors` (or sharded / `pytorch_model.bin`) -> state dicts.

The reference's inference scripts load the artefact its own fine-tune wrote, by PATH:
    /root/reference/AB/wavToWhisper.py:39,47      model_path = "./whisper-small-hi"; WhisperForConditionalGeneration.from_pretrained(model_path)
    /root/reference/AB/fineTuneMidiTester.py:20-21 model_dir = "./whisper-small-piano"; ...from_pretrained(model_dir)
    /root/reference/AB/fineTune.py:200             trainer.save_model()  -> <output_dir>/{config.json, model.safetensors | pytorch_model.bin, ...}
`finetune.WhisperLoRAModel.from_pretrained(path)` reads such a directory through this module; nothing here needs `transformers` or the
`safetensors` package (the format is an 8-byte little-endian header length, a JSON header {name: {dtype, shape, data_offsets}}, then the
raw little-endian tensor bytes).  `save_pretrained_dir` writes the same layout (used by the tests and by `Seq2SeqTrainer.save_model(full=True)`).

Key layout of `WhisperForConditionalGeneration.state_dict()` (HF:modeling_whisper.py:994-1010): `model.encoder.*`, `model.decoder.*`,
`proj_out.weight` (tied to `model.decoder.embed_tokens.weight`: safetensors files drop the duplicate, .bin files keep it).  `WhisperModel`
checkpoints have the same keys without the leading `model.`.
"""
from __future__ import annotations

import json
import os
import struct
from typing import Dict, Iterable, Optional, Tuple

import numpy as np
import torch

from .weights import EncoderConfig

_ST_DTYPES = {"F64": torch.float64, "F32": torch.float32, "F16": torch.float16, "BF16": torch.bfloat16, "I64": torch.int64, "I32": torch.int32,
              "I16": torch.int16, "I8": torch.int8, "U8": torch.uint8, "BOOL": torch.bool}
_ST_NAMES = {v: k for k, v in _ST_DTYPES.items()}


def read_safetensors(path: str, keys: Optional[Iterable[str]] = None) -> Dict[str, torch.Tensor]:
    """Tensors of a .safetensors file as CPU tensors in their stored dtype (memory-mapped, copied per tensor)."""
    with open(path, "rb") as f:
        head = f.read(8)
        if len(head) != 8:
            raise ValueError(f"{path}: not a safetensors file (shorter than its 8-byte header length)")
        (n,) = struct.unpack("<Q", head)
        size = os.path.getsize(path)
        if n <= 0 or 8 + n > size:
            raise ValueError(f"{path}: safetensors header length {n} does not fit the file ({size} bytes)")
        header = json.loads(f.read(n).decode("utf-8"))
    data = np.memmap(path, dtype=np.uint8, mode="r", offset=8 + n)
    want = None if keys is None else set(keys)
    out: Dict[str, torch.Tensor] = {}
    for name, meta in header.items():
        if name == "__metadata__" or (want is not None and name not in want):
            continue
        dt = _ST_DTYPES.get(meta["dtype"])
        if dt is None:
            raise ValueError(f"{path}: tensor {name} has unsupported dtype {meta['dtype']}")
        b, e = meta["data_offsets"]
        shape = tuple(int(s) for s in meta["shape"])
        count = int(np.prod(shape)) if shape else 1
        if e - b != count * torch.empty((), dtype=dt).element_size() or e > data.shape[0]:
            raise ValueError(f"{path}: tensor {name}: data_offsets {b}:{e} do not match shape {shape} / dtype {meta['dtype']}")
        buf = np.array(data[b:e])                                   # copy out of the mapping
        out[name] = torch.frombuffer(buf, dtype=dt, count=count).reshape(shape) if count else torch.empty(shape, dtype=dt)
    return out


def write_safetensors(path: str, tensors: Dict[str, torch.Tensor], metadata: Optional[Dict[str, str]] = None) -> None:
    header: Dict[str, object] = {}
    if metadata:
        header["__metadata__"] = metadata
    off, blobs = 0, []
    for name in sorted(tensors):
        t = tensors[name].detach().cpu().contiguous()
        raw = t.view(torch.uint8).numpy().tobytes() if t.numel() else b""
        header[name] = {"dtype": _ST_NAMES[t.dtype], "shape": list(t.shape), "data_offsets": [off, off + len(raw)]}
        off += len(raw)
        blobs.append(raw)
    hj = json.dumps(header, separators=(",", ":")).encode("utf-8")
    hj += b" " * ((8 - len(hj) % 8) % 8)                            # the reference writer pads the header to 8 bytes
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for raw in blobs:
            f.write(raw)


def _load_weight_files(path: str) -> Dict[str, torch.Tensor]:
    """Every tensor of the checkpoint directory, whichever of the layouts `save_pretrained` produces is present."""
    def have(name):
        return os.path.exists(os.path.join(path, name))

    sd: Dict[str, torch.Tensor] = {}
    if have("model.safetensors"):
        return read_safetensors(os.path.join(path, "model.safetensors"))
    if have("model.safetensors.index.json"):
        index = json.load(open(os.path.join(path, "model.safetensors.index.json")))
        for shard in sorted(set(index["weight_map"].values())):
            sd.update(read_safetensors(os.path.join(path, shard)))
        return sd
    if have("pytorch_model.bin"):
        return dict(torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True))
    if have("pytorch_model.bin.index.json"):
        index = json.load(open(os.path.join(path, "pytorch_model.bin.index.json")))
        for shard in sorted(set(index["weight_map"].values())):
            sd.update(torch.load(os.path.join(path, shard), map_location="cpu", weights_only=True))
        return sd
    raise FileNotFoundError(f"{path}: no model.safetensors, model.safetensors.index.json, pytorch_model.bin or pytorch_model.bin.index.json")


def load_checkpoint_dir(path: str) -> Tuple[dict, Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """(config.json as a dict, encoder state dict, decoder state dict) with the keys relative to the encoder / decoder modules, as fp32
    CPU tensors.  The tied `proj_out.weight` is checked against `decoder.embed_tokens.weight` when the file carries both."""
    path = os.fspath(path)
    if not os.path.isdir(path):
        raise FileNotFoundError(f"{path} is not a checkpoint directory (the reference loads `./whisper-small-hi`, AB/wavToWhisper.py:39)")
    cfg_path = os.path.join(path, "config.json")
    if not os.path.exists(cfg_path):
        raise FileNotFoundError(f"{cfg_path} is missing")
    cfg = json.load(open(cfg_path))
    raw = _load_weight_files(path)
    enc: Dict[str, torch.Tensor] = {}
    dec: Dict[str, torch.Tensor] = {}
    proj = None
    for k, v in raw.items():
        name = k[len("model."):] if k.startswith("model.") else k
        if name.startswith("encoder."):
            enc[name[len("encoder."):]] = v.float()
        elif name.startswith("decoder."):
            dec[name[len("decoder."):]] = v.float()
        elif name == "proj_out.weight":
            proj = v.float()
        else:
            raise KeyError(f"{path}: unexpected tensor {k} (expected model.encoder.*, model.decoder.*, proj_out.weight)")
    if not enc or not dec:
        raise KeyError(f"{path}: checkpoint has {len(enc)} encoder and {len(dec)} decoder tensors; both halves are required")
    if proj is not None and "embed_tokens.weight" in dec and not torch.equal(proj, dec["embed_tokens.weight"]):
        raise ValueError(f"{path}: proj_out.weight differs from decoder.embed_tokens.weight (Whisper ties them, HF:modeling_whisper.py:1003-1010)")
    if proj is not None and "embed_tokens.weight" not in dec:
        dec["embed_tokens.weight"] = proj
    return cfg, enc, dec


def encoder_config_from_hf(cfg: dict, name: str = "checkpoint") -> EncoderConfig:
    """`WhisperConfig` fields -> EncoderConfig (HF:configuration_whisper.py; defaults are the class defaults = Whisper-tiny)."""
    return EncoderConfig(int(cfg.get("d_model", 384)), int(cfg.get("encoder_layers", 4)), int(cfg.get("encoder_attention_heads", 6)),
                         int(cfg.get("encoder_ffn_dim", 1536)), int(cfg.get("num_mel_bins", 80)), int(cfg.get("max_source_positions", 1500)), name)


def save_pretrained_dir(path: str, cfg: dict, encoder_sd: Dict[str, torch.Tensor], decoder_sd: Dict[str, torch.Tensor], fmt: str = "safetensors",
                        dtype: torch.dtype = torch.float32) -> str:
    """Writes what `WhisperForConditionalGeneration.save_pretrained` writes for the two formats the reference's versions produce
    (`model.safetensors`: 4.53 / 5.x; `pytorch_model.bin`: 4.35's default), keys `model.encoder.*` / `model.decoder.*` (+ `proj_out.weight` in .bin)."""
    os.makedirs(path, exist_ok=True)
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"), indent=2, sort_keys=True)
    sd = {"model.encoder." + k: v.detach().cpu().to(dtype) for k, v in encoder_sd.items()}
    sd.update({"model.decoder." + k: v.detach().cpu().to(dtype) for k, v in decoder_sd.items()})
    if fmt == "safetensors":
        write_safetensors(os.path.join(path, "model.safetensors"), sd, {"format": "pt"})
    elif fmt == "bin":
        sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
        torch.save(sd, os.path.join(path, "pytorch_model.bin"))
    else:
        raise ValueError("fmt must be 'safetensors' or 'bin'")
    return path
