"""The `fineTune.py` call surface (/root/reference/AB/fineTune.py:131-200) over the native encoder.

Reference: `WhisperForConditionalGeneration.from_pretrained("openai/whisper-small")` is fully fine-tuned with HF
`Seq2SeqTrainer` (bs 16, lr 1e-5, warmup 1, 50 steps, fp32).  Build-defined variant asked for by BASELINE.json
configs[2]/[3]: the base weights are frozen, LoRA adapters on the encoder's q_proj / v_proj are the only trainable
parameters, and the step is data-parallel with one all-reduce of the adapter gradients.

What is native here and what is not:
* encoder forward + backward to the adapters: hand-written HIP via libawt (encoder.NativeWhisperEncoder, trainable=True);
* decoder (self-attention + cross-attention over the 1500 encoder positions + tied vocabulary projection) and the
  cross-entropy: stock PyTorch-ROCm ops below -- the decoder is the "next" row of the scope table, not part of the hot
  path, and it only has to carry d(loss)/d(encoder hidden states) back to the native backward.  Its one encoder-sized
  piece -- the cross-attention key / value projections of all 1500 encoder positions, 2 x layers GEMMs of
  [B*1500, d] x [d, d] -- runs on libawt's GEMM as ONE fused projection (`_CrossKVProjection`), forward and backward;
* AdamW on the adapter parameters: torch.optim.AdamW; learning-rate schedule: linear warm-up then linear decay, as HF's
  default `lr_scheduler_type="linear"`.

`Seq2SeqTrainingArguments` keeps the field names the reference passes (fineTune.py:162-183; `evaluation_strategy` and
`tokenizer=` are the 4.35-era spellings).  `evaluation_strategy="steps"` + `predict_with_generate` run `evaluate()` every
`eval_steps`: eval loss, greedy `generate` (self-attention cache, cross-attention keys / values from the fused native
projection) and the caller's `compute_metrics` (`wer()` below replaces `evaluate.load("wer")`), and
`load_best_model_at_end` restores the best adapters.  wandb / hub fields are accepted and ignored.  The decoder, the
loss and greedy decoding are pinned to `WhisperForConditionalGeneration` by tests/golden/decoder.npz.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Any, Callable, Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .dist import AwtComm, FlatGradBucket, world_size
from .encoder import NativeWhisperEncoder
from .weights import EncoderConfig, LoraSpec, unit_variates

WHISPER_VOCAB = 51865
DECODER_START = 50258   # <|startoftranscript|>
PAD_ID = 50257
EOS_ID = 50257          # <|endoftext|> (Whisper uses it as pad as well)


def shift_tokens_right(labels: torch.Tensor, pad_token_id: int, decoder_start_token_id: int) -> torch.Tensor:
    """HF:modeling_whisper.py:67-82."""
    shifted = labels.new_zeros(labels.shape)
    shifted[:, 1:] = labels[:, :-1].clone()
    shifted[:, 0] = decoder_start_token_id
    shifted.masked_fill_(shifted == -100, pad_token_id)
    return shifted


class _CrossKVProjection(torch.autograd.Function):
    """Cross-attention keys and values of every decoder layer, `[k_0 | v_0 | k_1 | ...] = enc W_all^T + b_all`, as one
    native GEMM ([B*S, d] x [d, 2*layers*d], HF:modeling_whisper.py:306-312 per layer), and d(enc) = dY W_all as one more.
    The projection weights are frozen, so there is no weight gradient."""

    @staticmethod
    def forward(ctx, enc2d, w_all, b_all, precision):
        ctx.save_for_backward(w_all)
        ctx.precision = precision
        return ops.linear(enc2d, w_all, b_all, precision)

    @staticmethod
    def backward(ctx, dy):
        (w_all,) = ctx.saved_tensors
        return ops.linear(dy.contiguous(), w_all.t().contiguous(), None, ctx.precision), None, None, None


class _Attention(nn.Module):
    """q/k/v/out projections under HF's names (`k_proj` has no bias, HF:modeling_whisper.py:262-282)."""

    def __init__(self, d: int, heads: int):
        super().__init__()
        self.heads = heads
        self.q_proj, self.k_proj = nn.Linear(d, d), nn.Linear(d, d, bias=False)
        self.v_proj, self.out_proj = nn.Linear(d, d), nn.Linear(d, d)

    def forward(self, x, kv_src=None, kv=None, causal=False, cache=None):
        """`kv`: precomputed (k, v) [B, S, d]; `cache`: dict holding the self-attention keys / values of earlier positions."""
        B, Lq, d = x.shape
        hd = d // self.heads
        if kv is None:
            src = x if kv_src is None else kv_src
            k, v = self.k_proj(src), self.v_proj(src)
            if cache is not None:
                if "k" in cache:
                    k, v = torch.cat([cache["k"], k], dim=1), torch.cat([cache["v"], v], dim=1)
                cache["k"], cache["v"] = k, v
        else:
            k, v = kv
        q, k, v = (t.reshape(B, -1, self.heads, hd).transpose(1, 2) for t in (self.q_proj(x), k, v))
        o = F.scaled_dot_product_attention(q, k, v, is_causal=causal and Lq > 1)
        return self.out_proj(o.transpose(1, 2).reshape(B, Lq, d))


class _DecoderLayer(nn.Module):
    """HF:modeling_whisper.py:416-507 (pre-LN; module names are the state-dict keys of `WhisperDecoderLayer`)."""

    def __init__(self, d: int, heads: int, ffn: int):
        super().__init__()
        self.self_attn, self.encoder_attn = _Attention(d, heads), _Attention(d, heads)
        self.self_attn_layer_norm, self.encoder_attn_layer_norm, self.final_layer_norm = nn.LayerNorm(d), nn.LayerNorm(d), nn.LayerNorm(d)
        self.fc1, self.fc2 = nn.Linear(d, ffn), nn.Linear(ffn, d)

    def forward(self, x, enc, kv=None, cache=None):
        x = x + self.self_attn(self.self_attn_layer_norm(x), causal=True, cache=cache)
        x = x + self.encoder_attn(self.encoder_attn_layer_norm(x), kv_src=enc, kv=kv)
        return x + self.fc2(F.gelu(self.fc1(self.final_layer_norm(x))))


class WhisperDecoder(nn.Module):
    """Pre-LN Whisper decoder with tied output projection (HF:modeling_whisper.py:416-507,649-797), stock torch ops.
    Parameter names are HF's (`WhisperDecoder.state_dict()` loads unchanged); pinned to the reference by
    tests/golden/decoder.npz."""

    def __init__(self, d: int, layers: int, heads: int, ffn: int, vocab: int = WHISPER_VOCAB, max_target_positions: int = 448):
        super().__init__()
        self.embed_tokens = nn.Embedding(vocab, d)
        self.embed_positions = nn.Embedding(max_target_positions, d)
        self.layers = nn.ModuleList([_DecoderLayer(d, heads, ffn) for _ in range(layers)])
        self.layer_norm = nn.LayerNorm(d)

    def cross_kv(self, enc: torch.Tensor, precision: str):
        """Per layer (k, v) [B, S, d] views of one fused native projection of the encoder output."""
        B, S, d = enc.shape
        att = [l.encoder_attn for l in self.layers]
        w_all = torch.cat([w for a in att for w in (a.k_proj.weight, a.v_proj.weight)], dim=0)
        b_all = torch.cat([b for a in att for b in (torch.zeros_like(a.v_proj.bias), a.v_proj.bias)], dim=0)
        with torch.autocast("cuda", enabled=False):
            kv = _CrossKVProjection.apply(enc.reshape(B * S, d).float(), w_all.float(), b_all.float(), precision)
        parts = kv.view(B, S, 2 * len(self.layers), d).unbind(dim=2)     # unbind: its backward is one stack, not 2L zero-fills
        return [(parts[2 * i], parts[2 * i + 1]) for i in range(len(self.layers))]

    def forward(self, input_ids: torch.Tensor, encoder_hidden_states: torch.Tensor, native_precision: Optional[str] = None,
                cross=None, caches=None, position_offset: int = 0) -> torch.Tensor:
        """`cross` / `caches` / `position_offset` serve incremental decoding (generate): precomputed cross-attention
        (k, v) per layer, per-layer self-attention caches, and the position of input_ids[:, 0]."""
        L = input_ids.shape[1]
        x = self.embed_tokens(input_ids) + self.embed_positions.weight[position_offset: position_offset + L]
        if cross is None:
            cross = self.cross_kv(encoder_hidden_states, native_precision) if native_precision else [None] * len(self.layers)
        for i, layer in enumerate(self.layers):
            x = layer(x, encoder_hidden_states, cross[i], None if caches is None else caches[i])
        return F.linear(self.layer_norm(x), self.embed_tokens.weight)   # proj_out tied to the embedding


class WhisperLoRAModel(nn.Module):
    """`model(input_features=..., labels=...) -> .loss, .logits` like WhisperForConditionalGeneration.forward
    (HF:modeling_whisper.py:994-1100): shift labels right, encoder, decoder, tied projection, CE with ignore -100."""

    def __init__(self, cfg: EncoderConfig, lora: Optional[LoraSpec], precision: Optional[str] = "bf16x3", device: str = "cuda", decoder_layers: Optional[int] = None,
                 seed: int = 0, vocab: int = WHISPER_VOCAB, decoder_autocast: Optional[torch.dtype] = None, native_cross_kv: bool = True,
                 max_target_positions: int = 448, backward_precision: Optional[str] = None, native_decoder: bool = True,
                 decoder_heads: Optional[int] = None, decoder_ffn: Optional[int] = None, decoder_lora: Optional[LoraSpec] = None):
        super().__init__()
        # decoder_lora: adapters on the decoder's self- / cross-attention q_proj, v_proj as well (scope row f1: "+ LoRA on decoder"; the
        # reference fine-tunes every decoder parameter, AB/fineTune.py:131,186-199) -- native decoder only
        if decoder_lora is not None and (not native_decoder or decoder_autocast is not None):
            raise ValueError("decoder_lora needs the native decoder (native_decoder=True, no decoder_autocast)")
        # lora=None: inference only (the reference's wavToWhisper.py / fineTuneMidiTester.py use): the encoder is not trainable and may
        # run any inference precision (precision=None picks it from the checkpoint, encoder.NativeWhisperEncoder)
        trainable = lora is not None
        # native_decoder=True (default): decoder, tied projection and cross-entropy run on libawt as well (native_decoder.NativeWhisperDecoder,
        # scope row f1); False keeps the stock-PyTorch decoder (the restatement pinned to HF by tests/golden/decoder.npz) around the
        # native encoder -- kept as the A/B reference of the native one (tests/test_gpu_native_decoder.py)
        if decoder_autocast is not None:
            native_decoder = False            # autocast is a property of the torch decoder
        self.native_decoder = native_decoder
        # the decoder is stock PyTorch (scope row "next"): fp32 like the reference (fp16=False, fineTune.py:170) unless
        # decoder_autocast=torch.bfloat16 asks torch to run its matmuls in bf16
        self.decoder_autocast = decoder_autocast
        self.native_cross_kv = native_cross_kv   # False: every decoder matmul on torch (the pre-fusion path, kept for A/B tests)
        self.encoder = NativeWhisperEncoder(cfg, precision=precision, lora=lora, device=device, seed=seed, trainable=trainable,
                                            backward_precision=backward_precision)
        self.precision = self.encoder.precision if precision is None else precision
        dheads, dffn = decoder_heads or cfg.heads, decoder_ffn or cfg.ffn
        torch.manual_seed(seed)
        self.decoder = WhisperDecoder(cfg.d_model, decoder_layers or cfg.layers, dheads, dffn, vocab, max_target_positions).to(device)
        if native_decoder:
            from .native_decoder import NativeWhisperDecoder
            nd = NativeWhisperDecoder(cfg.d_model, decoder_layers or cfg.layers, dheads, dffn, vocab, max_target_positions,
                                      precision=precision if precision in ("bf16", "bf16x3") else "bf16x3", lora=decoder_lora, lora_seed=seed).to(device)
            nd.load_state_dict(self.decoder.state_dict(), strict=decoder_lora is None)       # the same initial weights as the torch decoder of this seed
            self.decoder = nd
        for n, p in self.decoder.named_parameters():
            p.requires_grad = "lora_" in n      # frozen base model: only the adapters train
        self.config = SimpleNamespace(decoder_start_token_id=DECODER_START, pad_token_id=PAD_ID, eos_token_id=EOS_ID, d_model=cfg.d_model)

    def lora_parameters(self) -> List[nn.Parameter]:
        """Every trainable parameter: the encoder's adapters, then the decoder's (if any)."""
        return [p for n, p in self.encoder.named_parameters() if "lora_" in n] + [p for n, p in self.decoder.named_parameters() if "lora_" in n]

    @classmethod
    def from_pretrained(cls, path, lora: Optional[LoraSpec] = None, precision: Optional[str] = None, device: str = "cuda", **kw) -> "WhisperLoRAModel":
        """`WhisperForConditionalGeneration.from_pretrained(model_path)` for a LOCAL checkpoint directory, as the reference's inference
        scripts use it (AB/wavToWhisper.py:39,47 `./whisper-small-hi`; AB/fineTuneMidiTester.py:20-21 `./whisper-small-piano`): reads
        `config.json` and `model.safetensors` / `pytorch_model.bin` (checkpoint.load_checkpoint_dir; no `transformers`, no network), builds
        the model of that shape and loads encoder and decoder weights under their HF names.  lora=None: inference (precision None = chosen
        from the checkpoint: f16f8 unless it has outlier channels); a LoraSpec: adapters on the loaded (frozen) base for fine-tuning."""
        from .checkpoint import encoder_config_from_hf, load_checkpoint_dir
        hf, enc_sd, dec_sd = load_checkpoint_dir(path)
        cfg = encoder_config_from_hf(hf, os.path.basename(os.path.normpath(os.fspath(path))))
        if precision is None and lora is not None:
            precision = "bf16x3"
        model = cls(cfg, lora, precision=precision, device=device, decoder_layers=int(hf.get("decoder_layers", cfg.layers)),
                    vocab=int(hf.get("vocab_size", WHISPER_VOCAB)), max_target_positions=int(hf.get("max_target_positions", 448)),
                    decoder_heads=int(hf.get("decoder_attention_heads", cfg.heads)), decoder_ffn=int(hf.get("decoder_ffn_dim", cfg.ffn)), **kw)
        missing, unexpected = model.encoder.load_state_dict(enc_sd, strict=False)
        missing = [k for k in missing if "lora_" not in k]
        if missing or unexpected:
            raise KeyError(f"{path}: encoder state dict mismatch: missing {missing[:4]}, unexpected {list(unexpected)[:4]}")
        missing, unexpected = model.decoder.load_state_dict(dec_sd, strict=False)
        if [k for k in missing if "lora_" not in k] or unexpected:
            raise KeyError(f"{path}: decoder state dict mismatch: missing {[k for k in missing if 'lora_' not in k][:4]}, unexpected {list(unexpected)[:4]}")
        for n, p in model.decoder.named_parameters():
            p.requires_grad = "lora_" in n
        for key, attr in (("decoder_start_token_id", "decoder_start_token_id"), ("pad_token_id", "pad_token_id"), ("eos_token_id", "eos_token_id")):
            if hf.get(key) is not None:
                setattr(model.config, attr, int(hf[key]))
        model.name_or_path = os.fspath(path)
        return model

    def save_pretrained(self, path, fmt: str = "safetensors") -> str:
        """Writes the directory `from_pretrained` reads (and `WhisperForConditionalGeneration.from_pretrained` would): config.json + the base
        weights under `model.encoder.*` / `model.decoder.*`.  Adapters, if any, are MERGED into the projection weights (W + (alpha / r) B A),
        which is what a downstream loader of the directory expects to find there."""
        from .checkpoint import save_pretrained_dir
        enc = {k: v.detach().clone() for k, v in self.encoder.state_dict().items()}
        if self.encoder.lora is not None:
            scale = self.encoder.lora.alpha / self.encoder.lora.r
            for k in [k for k in enc if k.endswith(".lora_A")]:
                base = k[: -len(".lora_A")]
                enc[base + ".weight"] = enc[base + ".weight"] + scale * (enc[base + ".lora_B"] @ enc[k])
            enc = {k: v for k, v in enc.items() if "lora_" not in k}
        d = self.decoder
        dec = {k: v.detach().clone() for k, v in d.state_dict().items()}
        if getattr(d, "lora", None) is not None:
            for k in [k for k in dec if k.endswith(".lora_A")]:
                base = k[: -len(".lora_A")]
                dec[base + ".weight"] = dec[base + ".weight"] + d.lora.scale * (dec[base + ".lora_B"] @ dec[k])
            dec = {k: v for k, v in dec.items() if "lora_" not in k}
        cfg = {"architectures": ["WhisperForConditionalGeneration"], "model_type": "whisper", "d_model": self.encoder.cfg.d_model,
               "encoder_layers": self.encoder.cfg.layers, "encoder_attention_heads": self.encoder.cfg.heads, "encoder_ffn_dim": self.encoder.cfg.ffn,
               "num_mel_bins": self.encoder.cfg.n_mels, "max_source_positions": self.encoder.cfg.max_source_positions,
               "decoder_layers": len(d.layers), "decoder_attention_heads": d.layers[0].self_attn.heads if hasattr(d.layers[0].self_attn, "heads") else d.heads,
               "decoder_ffn_dim": d.layers[0].fc1.weight.shape[0], "vocab_size": d.embed_tokens.weight.shape[0],
               "max_target_positions": d.embed_positions.weight.shape[0], "decoder_start_token_id": self.config.decoder_start_token_id,
               "pad_token_id": self.config.pad_token_id, "eos_token_id": self.config.eos_token_id, "activation_function": "gelu",
               "scale_embedding": False, "torch_dtype": "float32"}
        return save_pretrained_dir(os.fspath(path), cfg, enc, dec, fmt=fmt)

    def forward(self, input_features: torch.Tensor, labels: Optional[torch.Tensor] = None, decoder_input_ids: Optional[torch.Tensor] = None):
        if decoder_input_ids is None:
            if labels is None:
                raise ValueError("either labels or decoder_input_ids is required")
            decoder_input_ids = shift_tokens_right(labels, self.config.pad_token_id, self.config.decoder_start_token_id)
        hidden = self.encoder(input_features).last_hidden_state
        if self.native_decoder:
            if labels is not None:
                loss, logits = self.decoder.loss(decoder_input_ids.to(hidden.device), labels.to(hidden.device), hidden)
            else:
                loss, logits = None, self.decoder(decoder_input_ids.to(hidden.device), hidden)
            return SimpleNamespace(loss=loss, logits=logits, encoder_last_hidden_state=hidden)
        with torch.autocast("cuda", dtype=self.decoder_autocast or torch.bfloat16, enabled=self.decoder_autocast is not None):
            logits = self.decoder(decoder_input_ids.to(hidden.device), hidden, self.precision if self.native_cross_kv else None)
        loss = None
        if labels is not None:
            loss = F.cross_entropy(logits.view(-1, logits.shape[-1]).float(), labels.to(hidden.device).reshape(-1), ignore_index=-100)
        return SimpleNamespace(loss=loss, logits=logits, encoder_last_hidden_state=hidden)

    @torch.no_grad()
    def generate(self, input_features: torch.Tensor, max_length: int = 225, eos_token_id: Optional[int] = None,
                 decoder_input_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Greedy decoding as `model.generate(input_features)` is used at AB/wavToWhisper.py:59 / fineTuneMidiTester.py:34 and by
        `predict_with_generate` (fineTune.py:172): encoder once, cross-attention keys / values once (the fused native
        projection), then one token per step with a self-attention cache.  Returns [B, <= max_length] token ids, starting
        with decoder_start_token_id; rows that have emitted `eos_token_id` are padded with pad_token_id."""
        hidden = self.encoder(input_features).last_hidden_state
        cross = self.decoder.cross_kv(hidden, self.precision) if (self.native_cross_kv or self.native_decoder) else None
        with torch.autocast("cuda", dtype=self.decoder_autocast or torch.bfloat16, enabled=self.decoder_autocast is not None):
            return greedy_decode(self.decoder, hidden, self.config.decoder_start_token_id, self.config.pad_token_id,
                                 self.config.eos_token_id if eos_token_id is None else eos_token_id, max_length, cross=cross,
                                 decoder_input_ids=decoder_input_ids)


@torch.no_grad()
def greedy_decode(dec: WhisperDecoder, hidden: torch.Tensor, start_id: int, pad_id: int, eos_id: int, max_length: int,
                  cross=None, decoder_input_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """argmax decoding with a self-attention cache; `cross` = per-layer (k, v) of `hidden` (computed here if None)."""
    B = hidden.shape[0]
    if cross is None:
        cross = [(l.encoder_attn.k_proj(hidden), l.encoder_attn.v_proj(hidden)) for l in dec.layers]
    ids = decoder_input_ids.to(hidden.device) if decoder_input_ids is not None else \
        torch.full((B, 1), start_id, dtype=torch.long, device=hidden.device)
    caches = [dict() for _ in dec.layers]
    done = torch.zeros(B, dtype=torch.bool, device=hidden.device)
    step_in, pos = ids, 0
    while ids.shape[1] < max_length:
        logits = dec(step_in, hidden, cross=cross, caches=caches, position_offset=pos)
        pos += step_in.shape[1]
        nxt = logits[:, -1].float().argmax(dim=-1)
        nxt = torch.where(done, torch.full_like(nxt, pad_id), nxt)
        ids = torch.cat([ids, nxt[:, None]], dim=1)
        done = done | (nxt == eos_id)
        if bool(done.all()):
            break
        step_in = nxt[:, None]
    return ids


def wer(references: List[str], predictions: List[str]) -> float:
    """Word error rate = word-level edit distance summed over the pairs / reference words: what `evaluate.load("wer")`
    (jiwer) returns at AB/fineTune.py:143-158 (the reference multiplies it by 100)."""
    errors = words = 0
    for ref, hyp in zip(references, predictions):
        r, h = ref.split(), hyp.split()
        prev = list(range(len(h) + 1))
        for i, rw in enumerate(r, 1):
            cur = [i] + [0] * len(h)
            for j, hw in enumerate(h, 1):
                cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (rw != hw))
            prev = cur
        errors += prev[len(h)]
        words += len(r)
    return errors / max(1, words)


@dataclass
class Seq2SeqTrainingArguments:
    output_dir: str = "./whisper-small-hi"
    per_device_train_batch_size: int = 16
    gradient_accumulation_steps: int = 1
    learning_rate: float = 1e-5
    warmup_steps: int = 1
    max_steps: int = 50
    gradient_checkpointing: bool = True      # accepted; activations fit in 288 GB, nothing is recomputed
    fp16: bool = False
    evaluation_strategy: str = "steps"
    per_device_eval_batch_size: int = 8
    predict_with_generate: bool = True
    generation_max_length: int = 225
    save_steps: int = 50
    eval_steps: int = 10
    logging_steps: int = 10
    report_to: List[str] = field(default_factory=list)
    load_best_model_at_end: bool = True
    metric_for_best_model: str = "wer"
    greater_is_better: bool = False
    push_to_hub: bool = False
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    seed: int = 42


def linear_schedule(step: int, warmup: int, total: int) -> float:
    """HF get_linear_schedule_with_warmup multiplier."""
    if step < warmup:
        return step / max(1, warmup)
    return max(0.0, (total - step) / max(1, total - warmup))


class Seq2SeqTrainer:
    def __init__(self, args: Seq2SeqTrainingArguments, model: WhisperLoRAModel, train_dataset=None, eval_dataset=None,
                 data_collator: Optional[Callable] = None, compute_metrics: Optional[Callable] = None, tokenizer: Any = None):
        self.args, self.model = args, model
        self.train_dataset, self.eval_dataset = train_dataset, eval_dataset
        self.data_collator, self.compute_metrics, self.tokenizer = data_collator, compute_metrics, tokenizer
        params = model.lora_parameters()
        self.optimizer = torch.optim.AdamW(params, lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2),
                                           eps=args.adam_epsilon, weight_decay=args.weight_decay)
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, lambda s: linear_schedule(s, args.warmup_steps, args.max_steps))
        # ONE flat gradient buffer: [ native encoder's adapters (library order; written by awt_encoder_backward_ex) | any
        # other trainable parameter ]; every .grad is a view of it, the exchange and the clipping run on it in place
        enc = getattr(model, "encoder", None)
        self.native = enc if hasattr(enc, "bind_grad_buffer") and getattr(enc, "trainable", False) else None
        if self.native is not None:
            first = self.native.lora_parameters_library_order()
            ids = {id(p) for p in first}
            rest = [p for p in params if id(p) not in ids]
            self.n_native = sum(p.numel() for p in first)
            flat = torch.zeros(self.n_native + sum(p.numel() for p in rest), dtype=torch.float32, device=self.native.device)
            self.native.bind_grad_buffer(flat[: self.n_native])
            self.bucket = FlatGradBucket(first + rest, flat=flat)
        else:
            self.n_native = 0
            self.bucket = FlatGradBucket(params)
        self.comm: Optional[AwtComm] = None          # libawt's RCCL communicator (GPU + "nccl" process group, or forced)
        self.exchange = "none"
        self._setup_exchange()
        self.log_history: List[Dict[str, float]] = []
        self.global_step = 0

    def _setup_exchange(self, force_native: bool = False) -> None:
        """Choose how the flat gradient buffer is averaged over the ranks: libawt's RCCL communicator, issued from inside the
        native backward on a side stream (GPU, "nccl" process group), or torch.distributed's all_reduce (gloo: CPU tests and
        single-GPU rehearsals).  `force_native` builds a communicator even for one rank (exercises the RCCL path on one GPU)."""
        import torch.distributed as dist
        world = world_size()
        backend = dist.get_backend() if world > 1 else None
        if self.native is not None and (force_native or (world > 1 and backend == "nccl")):
            self.comm = AwtComm(self.native.device)
            self.native.set_comm(self.comm, groups=2)
            self.exchange = "rccl (libawt awt_comm, in-backward, side stream, 2 layer groups)"
        elif world > 1:
            self.exchange = "torch.distributed all_reduce (%s) on the flat buffer" % backend

    def exchange_report(self):
        """What the last steps' gradient exchange did, for bench.py's multi-GPU lines: the transport, the ranks RCCL saw, and -- under libawt's
        communicator -- the side-stream time and size of every bucket reduction since the previous call (recording starts with the first call)."""
        rep = {"transport": self.exchange, "world": world_size(), "rccl_ranks": self.comm.world if self.comm is not None else 0,
               "flat_grad_elems": int(self.bucket.numel)}
        if self.comm is not None:
            spans = self.comm.bucket_stats(True)
            rep["buckets"] = [{"ms": round(ms, 4), "bytes": nb} for ms, nb in spans]
            rep["note"] = ("HIP events on the communicator's side stream around each ncclAllReduce(ncclAvg) of a finished layer group's adapter gradients, "
                           "issued inside the native backward; empty on the first call (recording starts with it)")
        return rep

    def exchange_checksums(self, R=None):
        """fp64 sum and sum of squares of this rank's flat gradient buffer AFTER the exchange (equal on every rank iff the all-reduce ran), gathered over
        the ranks when a process group exists: [[sum, sumsq], ...] by rank."""
        import torch.distributed as dist
        flat = self.bucket.flat.detach().double()
        mine = torch.stack([flat.sum(), (flat * flat).sum()])
        if dist.is_initialized() and dist.get_world_size() > 1:
            t = mine if dist.get_backend() == "nccl" else mine.cpu()
            out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
            dist.all_gather(out, t)
            return [[float(x[0]), float(x[1])] for x in out]
        return [[float(mine[0]), float(mine[1])]]

    def training_step(self, batch) -> float:
        """One optimizer step: forward + backward over `gradient_accumulation_steps` micro-batches (a single dict, or a list
        of that many dicts), ONE in-place mean all-reduce of the flat adapter-gradient buffer (inside the last native
        backward when libawt's communicator is attached), clipping on the flat buffer, AdamW, schedule."""
        dev = self.model.encoder.device
        micro = batch if isinstance(batch, (list, tuple)) else [batch]
        total = 0.0
        if self.native is not None:
            self.native.zero_adapter_grads()                # the first backward overwrites the bound buffer: no zero-fill kernel
            if self.bucket.numel > self.n_native:
                self.bucket.flat[self.n_native:].zero_()
        else:
            self.bucket.zero()
        for i, mb in enumerate(micro):
            if self.native is not None:
                self.native.grad_sync = i == len(micro) - 1   # exchange inside the last micro-batch's backward only
            out = self.model(input_features=mb["input_features"].to(dev), labels=mb["labels"].to(dev))
            (out.loss / len(micro)).backward()              # adapter .grad accumulates across micro-batches
            total += float(out.loss.detach())
        if self.comm is not None:
            self.bucket.allreduce_mean(comm=self.comm, segment=slice(self.n_native, None))   # the native segment is already averaged
        else:
            self.bucket.allreduce_mean()
        if self.args.max_grad_norm and self.args.max_grad_norm > 0:
            self.bucket.clip_norm_(self.args.max_grad_norm)
        self.optimizer.step()
        self.scheduler.step()
        self.global_step += 1
        return total / len(micro)

    def _batches(self):
        import torch.distributed as dist
        from .dist import shard_range
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        n = len(self.train_dataset)
        bs = self.args.per_device_train_batch_size
        g = torch.Generator().manual_seed(self.args.seed)
        while True:
            perm = torch.randperm(n, generator=g).tolist()
            for i in range(0, n, bs * world):
                lo, hi = shard_range(min(bs * world, n - i), rank, world)
                idx = perm[i + lo: i + hi]
                if idx:
                    yield self.data_collator([self.train_dataset[j] for j in idx])

    @torch.no_grad()
    def evaluate(self, eval_dataset=None) -> Dict[str, float]:
        """`predict_with_generate` evaluation (fineTune.py:172-181): loss on the eval set, greedy generation up to
        `generation_max_length`, then `compute_metrics(pred)` with `.predictions` / `.label_ids` as the reference's
        `compute_metrics` reads them (fineTune.py:145-158).  Keys are prefixed `eval_` like HF's."""
        ds = eval_dataset if eval_dataset is not None else self.eval_dataset
        if ds is None or self.data_collator is None:
            raise ValueError("evaluate() needs an eval_dataset and a data_collator")
        dev = self.model.encoder.device
        bs = self.args.per_device_eval_batch_size
        was_training = self.model.training
        self.model.eval()
        losses, weights, preds, labels = [], [], [], []
        for i in range(0, len(ds), bs):
            batch = self.data_collator([ds[j] for j in range(i, min(len(ds), i + bs))])
            feats, lab = batch["input_features"].to(dev), batch["labels"].to(dev)
            out = self.model(input_features=feats, labels=lab)
            n_tok = int((lab != -100).sum())
            losses.append(float(out.loss) * n_tok); weights.append(n_tok)
            if self.args.predict_with_generate:
                preds.append(self.model.generate(feats, max_length=self.args.generation_max_length).cpu())
                labels.append(lab.cpu())
        metrics = {"eval_loss": sum(losses) / max(1, sum(weights))}
        if self.args.predict_with_generate and self.compute_metrics is not None:
            width = max(p.shape[1] for p in preds)
            lw = max(l.shape[1] for l in labels)
            pad = self.model.config.pad_token_id
            pred = torch.cat([F.pad(p, (0, width - p.shape[1]), value=pad) for p in preds]).numpy()
            lab = torch.cat([F.pad(l, (0, lw - l.shape[1]), value=-100) for l in labels]).numpy()
            for k, v in self.compute_metrics(SimpleNamespace(predictions=pred, label_ids=lab)).items():
                metrics[k if k.startswith("eval_") else "eval_" + k] = float(v)
        self.model.train(was_training)
        metrics["step"] = self.global_step
        self.log_history.append(dict(metrics))
        return metrics

    def train(self):
        if self.args.report_to:
            print("[finetune] note: wandb / hub reporting is outside the native hot path and is skipped")
        do_eval = self.args.evaluation_strategy == "steps" and self.eval_dataset is not None and self.args.eval_steps > 0
        key = self.args.metric_for_best_model
        key = key if key.startswith("eval_") else "eval_" + key
        best, best_state = None, None
        it = self._batches()
        for _ in range(self.args.max_steps):
            loss = self.training_step([next(it) for _ in range(max(1, self.args.gradient_accumulation_steps))])
            if self.global_step % max(1, self.args.logging_steps) == 0 or self.global_step == 1:
                self.log_history.append({"step": self.global_step, "loss": loss, "lr": self.scheduler.get_last_lr()[0]})
            if do_eval and self.global_step % self.args.eval_steps == 0:
                m = self.evaluate()
                score = m.get(key, m["eval_loss"])
                better = best is None or (score > best if self.args.greater_is_better and key in m else score < best)
                if better:
                    best = score
                    best_state = ({k: v.detach().clone() for k, v in self.model.encoder.state_dict().items() if "lora_" in k},
                                  {k: v.detach().clone() for k, v in self.model.decoder.state_dict().items() if "lora_" in k})
            if self.args.save_steps and self.global_step % self.args.save_steps == 0:
                self.save_model()
        if self.args.load_best_model_at_end and best_state is not None:      # fineTune.py:178-180
            self.model.encoder.load_state_dict(best_state[0], strict=False)
            if best_state[1]:
                self.model.decoder.load_state_dict(best_state[1], strict=False)
        train_losses = [h["loss"] for h in self.log_history if "loss" in h]
        return SimpleNamespace(global_step=self.global_step, training_loss=train_losses[-1] if train_losses else float("nan"),
                               best_metric=best)

    def save_model(self, output_dir: Optional[str] = None, full: bool = False):
        """Adapter-only checkpoint (the frozen base is not rewritten): `<output_dir>/lora_adapters.pt`, HF-style keys.  full=True also writes
        what the reference's `trainer.save_model()` leaves behind (AB/fineTune.py:200): a checkpoint DIRECTORY (config.json + model.safetensors,
        adapters merged into the base) that `WhisperLoRAModel.from_pretrained` -- and the reference's wavToWhisper.py:47 -- load by path."""
        out = output_dir or self.args.output_dir
        os.makedirs(out, exist_ok=True)
        if full:
            self.model.save_pretrained(out)
        sd = {k: v.detach().cpu() for k, v in self.model.encoder.state_dict().items() if "lora_" in k}
        sd.update({"decoder." + k: v.detach().cpu() for k, v in self.model.decoder.state_dict().items() if "lora_" in k})
        torch.save({"lora": sd, "r": self.model.encoder.lora.r, "alpha": self.model.encoder.lora.alpha,
                    "targets": list(self.model.encoder.lora.targets)}, os.path.join(out, "lora_adapters.pt"))
        return out
