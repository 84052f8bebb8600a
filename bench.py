#!/usr/bin/env python3
"""Throughput of the hot path: 4 s @ 16 kHz clips/sec through log-mel + Whisper-small encoder on N MI355X.

    python bench.py --gpus N --steps K --warmup W          # N > 1: this process starts N ranks itself (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic clips per GPU: int16 PCM [B, 64000] already
resident in HBM -> Whisper log-mel [B, 80, 3000] -> Whisper-small encoder -> last_hidden_state [B, 1500, 768] fp32,
all inside `awt_audio_encode` (libawt, hand-written HIP).  Reference semantics ("parity mode"): every clip is
zero-padded to 30 s and all 1500 positions are attended; weights are random-init of the Whisper-small
architecture (no checkpoint offline).  Clips shard across ranks with no data-path collective (weak scaling).

Workloads (--workload):
  encode    (default, BASELINE.json's metric) the step above, K times over the same resident batch
  sweep     BASELINE.json configs[4]: the seeded 10 000-clip piano-note set, pre-staged per rank as one int16 device tensor
            (contiguous shard, dist.shard_range), walked once in batches of B incl. the short tail batch
  finetune  configs[2]/[3]: one LoRA fine-tune step = log-mel + encoder fwd/bwd + decoder/CE + gradient exchange + AdamW
  noop      launcher / rendezvous / timing / JSON rehearsal without touching the GPU (CPU test of the N > 1 plumbing)

The JSON line carries, besides the driver's contract fields:
  roofline      the dominant kernel class (the MFMA GEMMs), timed live with HIP events on the launch stream;
                achieved = algorithmic FLOP (2 M N K per launch) / event time, peak = dense bf16 MFMA peak
  cpu_baseline  oracle/ (CPU restatement of the reference) timed on this host's cores on a bounded sample (rank 0, N=1)
  parity        HIP outputs vs that oracle on the same sample
  ranks         per-rank clips/s, the collective backend and the world size an all-reduce of ones actually returned
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
T_START = time.time()      # process start: `wall_s` in the JSON line is the whole run as the driver's clock sees it (weights, warm-up, CPU baseline, side blocks)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_DENSE_TFLOPS = 2500.0   # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
# operand + result bytes of the 50 GEMM launches of one step (Whisper-small, B = 64, two 2-byte operand planes, fp32 residual):
# per layer qkv (295 + 885 MB), out (295 + 590), fc1 (295 + 1180), fc2 (1180 + 590) = 5.31 GB; conv stem 0.2 + 1.2 + 0.3 GB
GEMM_ALGO_BYTES_PER_STEP = 12 * 5.31e9 + 1.7e9
# the same with fp16-exact weights in the f16f8 format (3-byte operands: fp16 + lo8 activations, fp16 + hi8 weights; v without e4m3 images):
# per layer qkv (221 + 5 + 737 MB), out (221 + 2 + 590), fc1 (221 + 7 + 885), fc2 (885 + 7 + 590) = 4.37 GB
GEMM_ALGO_BYTES_PER_STEP_EXACT = 12 * 4.37e9 + 1.7e9
ENCODER_GFLOP_PER_CLIP = {("small", False): 344.16, ("small", True): 36.30, ("tiny", False): 36.94, ("tiny", True): 3.33,
                          ("base", False): 87.37}
PRECISIONS = ["f16f8", "fp16x3", "bf16x3", "bf16", "fp16"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="small")
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU per step")
    ap.add_argument("--precision", default=None, choices=PRECISIONS,
                    help="operand precision (encoder.PRECISIONS).  Default: f16f8 (fp16 main product + block-scaled e4m3 cross terms, two "
                         "MFMA-equivalents per fragment pair) for the encode / sweep workloads, bf16x3 for finetune.  fp16x3 and bf16x3 are the "
                         "three-product split modes; bf16 is the single-pass mode that misses the 1e-3 bound")
    ap.add_argument("--trimmed", action="store_true", help="T=400/S=200 mode (NOT reference-equivalent)")
    ap.add_argument("--chunk", type=int, default=0, help="clips per kernel wave inside the library (0 = default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-mode", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="encode workload, 1 GPU: do not append the compact blocks of the other 1-GPU configurations "
                                                              "(Whisper-tiny B=32, the fine-tune step, the 10k-clip sweep)")
    ap.add_argument("--weights", default="fp32", choices=["fp16", "fp32"],
                    help="fp32 (default, what SURVEY.md 8(d) specifies and what the reference's fully fp32-fine-tuned checkpoints hold): the seed-0 "
                         "N(0, 0.02^2) initialisation as arbitrary fp32 values.  fp16: the same values rounded to fp16-representable ones (a frozen "
                         "base released in half precision); the library detects that at upload and its f16f8 GEMMs drop the then-zero x_hi w_lo "
                         "cross term (DESIGN.md 4.2b).  The other case is measured in the same run and reported beside the headline with its own roofline")
    ap.add_argument("--no-power", action="store_true", help="do not sample package power / shader clock from sysfs")
    ap.add_argument("--cpu-clips", type=int, default=8)
    ap.add_argument("--workload", default="encode", choices=["encode", "sweep", "finetune", "noop"])
    ap.add_argument("--clips", type=int, default=10000, help="sweep workload: clips in the whole set (sharded over the ranks)")
    ap.add_argument("--lora-r", type=int, default=8)
    ap.add_argument("--decoder-dtype", default="fp32", choices=["fp32", "bf16"], help="finetune workload with --torch-decoder: dtype of the stock-PyTorch decoder")
    ap.add_argument("--torch-decoder", action="store_true", help="finetune workload: stock-PyTorch decoder + CE instead of the native ones (A/B)")
    ap.add_argument("--backward-precision", default=None, choices=["bf16", "f16f8", "bf16x3"],
                    help="finetune workload.  Default f16f8: the MLP of the step (fc1 / fc2 forward and their two backward GEMMs) in the f16f8 operand format, "
                         "everything else in the forward's split-bf16 (same gradient error against the oracle's autograd, DESIGN.md 4.5); bf16x3: split-bf16 "
                         "everywhere (the library's own default); bf16: gradient contractions in single bf16 products (~0.5 %% gradient error)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the real multi-GPU path) or gloo (rehearsal of the N > 1 code path on one GPU: every rank on cuda:0)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus() -> int:
    """GPUs this process could open, counted WITHOUT touching HIP or importing torch (the launcher parent must stay a process that has
    never initialised the GPU): KFD topology nodes with SIMDs, capped by the DRM render nodes present and by HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES.  The rank processes check again with torch (Ranks.__init__), which is the authoritative answer."""
    import glob
    n = 0
    for props in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(props):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
        except OSError:
            pass
    render = len(glob.glob("/dev/dri/renderD*"))
    if render:
        n = min(n, render)
    if n == 0 and render and not os.path.isdir("/sys/class/kfd/kfd/topology/nodes"):
        n = render       # a container without the KFD topology in sysfs: the DRM render nodes it was given are the count; still no torch, no HIP here
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(a) -> int:
    """`python bench.py --gpus N` without a torchrun wrapper: start N fresh rank processes (one per GPU), wait for them, and return the
    worst exit code.  This parent imports neither torch nor anything that could initialise HIP (a GPU-initialised process must not start
    other programs on this pool); the visible-GPU check here reads sysfs only and every rank repeats it with torch (exit 2).  Rank 0's
    stdout (the JSON line) is this process'."""
    if a.dist_backend == "nccl" and a.workload != "noop":
        have = visible_gpus()
        if have < a.gpus:
            print(f"bench.py: --gpus {a.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
            return 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(a.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # poll every rank: the first one to fail (or the deadline) ends the others -- a dead rank leaves the rest inside a collective
    worst, deadline = 0, time.time() + 3300
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc and not worst:
                worst = rc
        if live and (worst or time.time() > deadline):
            for q in live:
                q.kill()
            for q in live:
                q.wait()
            worst = worst or 124
            break
        if live:
            time.sleep(0.05)
    return worst


class Ranks:
    """RANK / LOCAL_RANK / WORLD_SIZE plumbing shared by every workload: process group, barrier + max-over-ranks timing,
    per-rank gathers, and a check that the group really spans `--gpus` ranks."""

    def __init__(self, a, need_gpu: bool = True):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={self.world}")
        self.backend = a.dist_backend if self.world > 1 else "none"
        self.dev = None
        if need_gpu:
            if a.dist_backend == "nccl" and torch.cuda.device_count() < self.world:      # device_count() does not initialise the GPU on this image
                print(f"bench.py: --gpus {self.world} but only {torch.cuda.device_count()} GPU(s) are visible", file=sys.stderr)
                raise SystemExit(2)
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
            self.dev = torch.device("cuda", self.local_rank if a.dist_backend == "nccl" else 0)
            torch.cuda.set_device(self.dev)
        self.collective_ranks = 1
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if a.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group("gloo")
            ones = torch.ones(1, dtype=torch.float64, device=self.dev if (need_gpu and a.dist_backend == "nccl") else "cpu")
            dist.all_reduce(ones)
            self.collective_ranks = int(ones.item())
            if self.collective_ranks != a.gpus:
                raise SystemExit(f"the {a.dist_backend} group spans {self.collective_ranks} ranks, --gpus asked for {a.gpus}")

    def sync(self):
        if self.dev is not None:
            self.torch.cuda.synchronize(self.dev)

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def timed(self, fn, steps: int):
        """barrier + synchronize on both sides of exactly `steps` calls of fn; returns (max-over-ranks seconds, this rank's
        seconds, last result)."""
        self.sync(); self.barrier(); self.sync()
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = fn()
        self.sync()
        mine = time.perf_counter() - t0
        self.barrier()
        dt = time.perf_counter() - t0
        return self.max(dt), mine, out

    def _cpu_or_dev(self):
        return self.dev if (self.dev is not None and self.backend == "nccl") else "cpu"

    def max(self, x: float) -> float:
        if self.world == 1:
            return x
        import torch.distributed as dist
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self._cpu_or_dev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, x: float):
        """[x of rank 0, x of rank 1, ...] on every rank."""
        if self.world == 1:
            return [x]
        import torch.distributed as dist
        t = self.torch.zeros(self.world, dtype=self.torch.float64, device=self._cpu_or_dev())
        t[self.rank] = x
        dist.all_reduce(t)
        return [float(v) for v in t.tolist()]

    def describe(self, per_rank_units_per_s):
        return {"world": self.world, "backend": {"nccl": "rccl (torch.distributed 'nccl')", "gloo": "gloo", "none": "none"}[self.backend],
                "collective_ranks": self.collective_ranks, "rccl_ranks": self.collective_ranks if self.backend == "nccl" else 0,
                "per_rank_clips_per_s": [round(v, 2) for v in per_rank_units_per_s]}

    def finish(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()


def source_hash() -> str:
    """sha256 over the HIP sources + headers: ties a committed rocprof summary to the build it was measured on."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "mlx8-ws-audio-transformer_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode()); h.update(open(os.path.join(csrc, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "awt.h"), "rb").read())
    return h.hexdigest()[:16]


def device_info(torch, dev):
    props = torch.cuda.get_device_properties(dev)
    return {"name": props.name, "arch": getattr(props, "gcnArchName", ""), "compute_units": props.multi_processor_count,
            "hbm_gib": round(props.total_memory / 2 ** 30, 1),
            "note": "peaks used: 2.5 PFLOP/s dense bf16 MFMA, 8 TB/s HBM3E (MI355X_MICROARCH.md); the encoder runs at the "
                    "1400 W package limit with sclk 1.9-2.1 GHz (DESIGN.md 4.2)"}


# ------------------------------------------------------------------------------------------------ workloads
def noop_main(a):
    """No GPU: the launcher, the rendezvous, the barrier / max-over-ranks timing and the JSON assembly of the real workloads."""
    R = Ranks(a, need_gpu=False)
    dt, mine, _ = R.timed(lambda: time.sleep(0.001), a.steps)
    rates = R.gather(a.batch * a.steps / mine)
    if R.rank == 0:
        print(json.dumps({"metric": "launcher rehearsal (no GPU work)", "value": round(a.batch * R.world * a.steps / dt, 2), "unit": "clips/s",
                          "n_gpus": R.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none",
                          "config": {"workload": "noop"}, "ranks": R.describe(rates)}))
    R.finish()


def finetune_cpu_baseline(a, torch, cfg, labels_all):
    """The fine-tune step's CPU baseline: the oracle encoder (with the LoRA term) + the stock-PyTorch restatement of the decoder / loss
    (finetune.WhisperDecoder, pinned to WhisperForConditionalGeneration by tests/golden/decoder.npz) under torch autograd, forward + backward
    to the adapters, on 2 of the step's clips (1 warm-up + 2 timed passes, median) on this process' CPU share (cpu_budget())."""
    import numpy as np
    import torch.nn.functional as F
    from mlx8_ws_audio_transformer_amd import synth, weights as wts
    from mlx8_ws_audio_transformer_amd.finetune import WhisperDecoder, shift_tokens_right, DECODER_START, PAD_ID
    from oracle import encoder as oenc, logmel as omel
    threads, note = cpu_budget()
    torch.set_num_threads(threads)
    n = 2
    spec = wts.LoraSpec(r=a.lora_r, alpha=16.0)
    W = wts.init_encoder_weights(cfg, 0, "hf")
    lora = {k: torch.from_numpy(v).requires_grad_(True) for k, v in wts.init_lora_weights(cfg, spec, 0, zero_b=False).items()}
    torch.manual_seed(0)
    dec = WhisperDecoder(cfg.d_model, cfg.layers, cfg.heads, cfg.ffn)
    for p_ in dec.parameters():
        p_.requires_grad = False
    clips = [synth.pcm_i16_to_f32(c) for c in synth.synth_clips_i16(n, seed=1234, first=0)]
    labels = labels_all[:n].clone()
    times = []
    for it in range(3):
        t0 = time.perf_counter()
        mel = omel.whisper_logmel(clips, n_samples=cfg.n_frames * 160)
        hidden = oenc.encoder_forward({**{k: torch.from_numpy(v) for k, v in W.items()}, **lora}, mel, cfg.heads, lora_scale=spec.scale)
        logits = dec(shift_tokens_right(labels, PAD_ID, DECODER_START), hidden)
        loss = F.cross_entropy(logits.view(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=-100)
        for v in lora.values():
            v.grad = None
        loss.backward()
        if it:
            times.append(time.perf_counter() - t0)
    m = sorted(times)[len(times) // 2]
    return {"value": round(n / m, 3), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d of the step's clips: oracle log-mel + encoder with LoRA r=%d + torch decoder + CE, forward and backward under autograd (no optimizer); "
                      "1 warm-up + 2 timed passes, median %.2f s per pass" % (n, a.lora_r, m), "threads_note": note}


def finetune_measure(a, R, bwd, with_cpu_baseline):
    """One measured fine-tune configuration on this rank's GPU (see finetune_main): warm-up, EXACTLY a.steps timed steps, class shares.
    bwd: None = the library default (every GEMM of the step in the forward's split-bf16 format), "f16f8" = the MLP's four GEMMs per layer in f16f8."""
    torch, dev, rank, world = R.torch, R.dev, R.rank, R.world
    from mlx8_ws_audio_transformer_amd import _lib, synth, weights as wts
    from mlx8_ws_audio_transformer_amd.feature_extraction import logmel_whisper_device
    from mlx8_ws_audio_transformer_amd.finetune import Seq2SeqTrainer, Seq2SeqTrainingArguments, WhisperLoRAModel
    cfg = wts.config(a.model, a.trimmed)
    B = a.batch
    pcm = torch.from_numpy(synth.synth_clips_i16(B, seed=1234, first=rank * B)).to(dev)
    model = WhisperLoRAModel(cfg, wts.LoraSpec(r=a.lora_r, alpha=16.0), precision=a.precision, device=str(dev),
                             decoder_autocast=torch.bfloat16 if (a.decoder_dtype == "bf16" and a.torch_decoder) else None,
                             backward_precision=bwd, native_decoder=not a.torch_decoder)
    g = torch.Generator().manual_seed(rank)
    labels = torch.randint(0, 51864, (B, 12), generator=g); labels[:, 0] = 50258
    args = Seq2SeqTrainingArguments(per_device_train_batch_size=B, learning_rate=1e-5, max_steps=10 ** 6, predict_with_generate=False)
    tr = Seq2SeqTrainer(args=args, model=model)
    state = {}

    def step():
        feats = logmel_whisper_device(pcm, n_frames=cfg.n_frames)       # mel is part of the step, as in the encode workload
        state["loss"] = tr.training_step({"input_features": feats, "labels": labels})

    for _ in range(a.warmup):
        step()
    # the dominant kernel class (every GEMM of the step: encoder forward + backward, adapters, decoder) is event-timed live in the timed
    # region; the attention classes' shares come from one extra untimed step
    _lib.prof_enable(True, ["gemm"])
    for k in _lib.PROF_CLASSES:
        _lib.prof_collect(k)
    dt, mine, _ = R.timed(step, a.steps)
    gemm_ms, gemm_n, gemm_flop = _lib.prof_collect("gemm")
    _lib.prof_enable(True, [k for k in _lib.PROF_CLASSES if k != "gemm"])
    tr.exchange_report()                                   # starts (and clears) the bucket timing of libawt's communicator: the extra step below is the one reported
    step()
    shares = {"gemm": round(gemm_ms / a.steps, 3)}
    other = {}
    for k in _lib.PROF_CLASSES:
        if k != "gemm":
            other[k] = _lib.prof_collect(k)
            shares[k] = round(other[k][0], 3)
    _lib.prof_enable(False)
    exchange = tr.exchange_report()
    rates = R.gather(B * a.steps / mine)
    result = None
    if rank == 0:
        achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        terms = float(_lib.MFMA_PER_PAIR[a.precision])
        bwd_ms, bwd_n, bwd_flop = other.get("attention_bwd", (0.0, 0, 0.0))
        result = {
            "metric": "4s@16kHz clips/sec through mel+Whisper-%s LoRA fine-tune step" % a.model, "value": round(B * world * a.steps / dt, 2),
            "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": arithmetic_dtype(a.precision), "data": "synthetic",
            "device": device_info(torch, dev),
            "config": {"workload": "LoRA r=%d (q_proj, v_proj) fine-tune step: log-mel + encoder fwd/bwd (HIP) + decoder/CE (%s) + gradient exchange + AdamW" % (
                           a.lora_r, "torch" if a.torch_decoder else "HIP"),
                       "clips_per_gpu_per_step": B, "global_batch": B * world, "precision": a.precision, "label_tokens": 12, "decoder_dtype": a.decoder_dtype,
                       "backward_precision": bwd or a.precision,
                       "mlp_operand_format": ("f16f8: fc1 / fc2 forward and their two backward GEMMs on the fp16 + 2 x e4m3 operand format (2 MFMA-equivalents, 2^-16 per "
                                              "operand), gradients carried x 2^k (k from max |d loss / d hidden| each step)" if bwd == "f16f8" else "as the rest of the step"),
                       "decoder_cross_attention": ("torch" if a.torch_decoder else
                                                   "absorbed (q_h W_k,h against the encoder states, batched GEMMs; native_decoder._AbsorbedCross)"
                                                   if getattr(model.decoder, "absorbed_cross", lambda *_: False)(12, cfg.max_source_positions)
                                                   else "projected keys / values (cross_kv)"),
                       "weights": "seed-0 random init, arbitrary fp32 values (SURVEY.md 8(d) C3); adapters A ~ N(0, 1/d), B = 0",
                       "adapter_grad_elems": tr.bucket.numel, "gradient_exchange": tr.exchange,
                       "parallelism": "dp%d, one in-place mean all-reduce of %.2f MB per step" % (world, tr.bucket.numel * 4 / 1e6)},
            "roofline": {"bound": "mfma", "kernel": "gemm_kernel<%s>%s (every GEMM of the step: encoder forward and backward, adapter terms, decoder)" % (
                             a.precision, " + gemm_f8_kernel (the MLP's four GEMMs per layer)" if bwd == "f16f8" else ""),
                         "achieved": round(achieved, 2), "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_DENSE_TFLOPS, 4),
                         "traffic": None, "launches": gemm_n, "avg_launch_ms": round(gemm_ms / max(gemm_n, 1), 4),
                         "mfma_issue_frac": round(terms * achieved / PEAK_BF16_DENSE_TFLOPS, 4),
                         "note": "algorithmic FLOP = 2 M N K per launch; HIP events on the launch stream inside the timed region"},
            "roofline_attention_backward": {
                "bound": "mfma", "kernel": "attention_bwd_kernel (dq launch + dk/dv launch per layer)",
                "achieved": round(bwd_flop * (10.0 / 14.0) / (bwd_ms * 1e-3) / 1e12, 2) if bwd_ms > 0 else None, "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                "frac": round(bwd_flop * (10.0 / 14.0) / (bwd_ms * 1e-3) / 1e12 / PEAK_BF16_DENSE_TFLOPS, 4) if bwd_ms > 0 else None,
                "launches": bwd_n, "ms_per_step": round(bwd_ms, 3),
                "note": "algorithmic FLOP = five S x S x 64 products per head (s, dv, dp, dq, dk) = 10 B H S^2 64; the two launches form s and dp twice (seven products)"},
            "time_share_ms_per_step": shares,
            "ranks": R.describe(rates), "last_loss": state["loss"], "build": source_hash()}
        result["gradient_exchange"] = exchange
        tpath = os.path.join(ROOT, "profiles", "traffic_finetune.json")      # rocprofv3 --pmc passes of this workload (tools/profile_round.sh), keyed by build
        if os.path.exists(tpath) and a.model == "small" and B == 64:
            try:
                tj = json.load(open(tpath))
                hit = [e for e in tj.get("entries", [tj]) if e.get("build") == source_hash() and e.get("backward_precision") == (bwd or a.precision)]
                if hit:
                    result["roofline"]["traffic"] = round(hit[-1]["per_kernel"]["gemm_kernel"]["hbm_bytes_per_launch"])
                    result["roofline"]["traffic_note"] = "bytes per GEMM launch, FETCH_SIZE x2 (gfx950) + WRITE_SIZE, rocprofv3 --pmc passes on this build (profiles/traffic_finetune.json)"
            except Exception:
                pass
        if world == 1 and with_cpu_baseline:
            result["cpu_baseline"] = finetune_cpu_baseline(a, torch, cfg, labels)
    sums = tr.exchange_checksums()                         # collective: [sum, sum of squares] of every rank's flat gradient buffer after the exchange
    if rank == 0:
        result["gradient_checksum_per_rank"] = sums
        result["gradient_checksums_equal"] = all(abs(x[0] - sums[0][0]) <= 1e-9 * max(1.0, abs(sums[0][0])) and abs(x[1] - sums[0][1]) <= 1e-9 * max(1.0, abs(sums[0][1])) for x in sums)
    del tr, model
    torch.cuda.empty_cache()
    return result


def finetune_main(a):
    """BASELINE.json configs[2]/[3]: Whisper-small + LoRA (q_proj, v_proj) fine-tune step, B clips per GPU, 12 label tokens,
    one in-place mean all-reduce of the flat adapter-gradient buffer per step (RCCL through libawt's communicator under the
    nccl backend).  Encoder forward / backward, decoder and loss are libawt kernels; AdamW on the adapters is torch.optim.
    `value` is the LIBRARY DEFAULT (every GEMM in split-bf16, documented gradient error 3e-5); the step with the MLP's GEMMs in f16f8
    (gradient error <= 2e-3 vs the oracle, tests/test_gpu_backward.py) is reported beside it as `f16f8_mlp` unless --backward-precision picks one."""
    R = Ranks(a)
    if a.backward_precision is not None:
        bwd = None if a.backward_precision == "bf16x3" else a.backward_precision
        result = finetune_measure(a, R, bwd, not a.no_cpu_baseline)
    else:
        result = finetune_measure(a, R, None, not a.no_cpu_baseline)
        if a.precision == "bf16x3":
            for key, bwd in (("f16f8_mlp", "f16f8"), ("bf16_backward", "bf16")):
                fast = finetune_measure(a, R, bwd, False)
                if R.rank == 0:
                    result[key] = {k: fast[k] for k in ("value", "unit", "ms_per_step", "roofline", "roofline_attention_backward", "time_share_ms_per_step", "last_loss")}
                    result[key]["config"] = {k: fast["config"][k] for k in ("backward_precision", "mlp_operand_format")}
            if R.rank == 0:
                result["bf16_backward"]["note"] = ("mixed-precision backward (awt_encoder_cfg.backward_terms = 1): the backward GEMMs and the attention backward's dp / dq / dk / dv use ONE "
                                                   "bf16 product per fragment pair, scores recomputed in split-bf16, forward untouched; adapter gradients within 2e-2 rel-L2 of the default "
                                                   "(tests/test_gpu_backward.py::test_bf16_backward_option_stays_close_to_the_exact_backward) -- reported beside, never as, `value`")
    if R.rank == 0:
        result["wall_s"] = round(time.time() - T_START, 1)
        print(json.dumps(result))
    R.finish()


def sweep_measure(a, R, shard, t_synth, with_cpu_baseline):
    """The whole seeded clip set, end to end, once, on this rank's pre-staged shard (host int16 [n_local, 64000])."""
    from mlx8_ws_audio_transformer_amd import _lib, sweep, weights as wts
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    torch, dev = R.torch, R.dev
    cfg = wts.config(a.model, a.trimmed)
    pcm = torch.from_numpy(shard).to(dev)
    enc = load_bench_weights(torch, NativeWhisperEncoder(cfg, precision=a.precision, device=str(dev), chunk_clips=a.chunk, seed=0, init_profile="hf").eval(),
                             bench_weights(cfg, a.weights))
    for _ in range(max(1, a.warmup)):
        enc.encode_pcm(pcm[: a.batch])
    check = torch.zeros((), dtype=torch.float64, device=dev)

    def sink(b0, hidden):
        check.add_(hidden.sum(dtype=torch.float64))     # every hidden state is consumed once: the batches are really computed

    _lib.prof_enable(True, ["gemm"])
    _lib.prof_collect("gemm")
    dt, mine, n_local = R.timed(lambda: sweep.encode_sweep(enc, pcm, a.batch, sink), 1)
    gemm_ms, gemm_n, gemm_flop = _lib.prof_collect("gemm")
    _lib.prof_enable(False)
    rates = R.gather(n_local / mine)
    checks = R.gather(float(check.item()))
    nb = -(-pcm.shape[0] // a.batch)
    result = None
    if R.rank == 0:
        value = a.clips / dt
        gf = ENCODER_GFLOP_PER_CLIP.get((a.model, a.trimmed))
        achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        result = {
            "metric": "4s@16kHz clips/sec through mel+Whisper-%s encoder, %d-clip sweep" % (a.model, a.clips), "value": round(value, 2),
            "unit": "clips/s", "n_gpus": R.world, "steps": nb, "warmup": max(1, a.warmup), "ms_per_step": round(dt / nb * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": arithmetic_dtype(a.precision), "data": "synthetic", "device": device_info(torch, dev),
            "config": {"workload": "%d seeded piano-note clips (synth.py, seed 1234; distributions of AB/synthDataset.py:46-76) pre-staged as int16 in HBM, "
                                   "contiguous shard per rank, batches of %d incl. the tail batch -> log-mel -> Whisper-%s encoder -> hidden states" % (a.clips, a.batch, a.model),
                       "mode": "trimmed (NOT reference-equivalent)" if a.trimmed else "parity (clip zero-padded to 30 s, as the reference computes)",
                       "clips_total": a.clips, "clips_this_rank": int(pcm.shape[0]), "batches_this_rank": nb, "tail_batch": int(pcm.shape[0] % a.batch),
                       "precision": a.precision, "weights": a.weights, "parallelism": "dp%d (contiguous clip shards, no data-path collective)" % R.world},
            "roofline": {"bound": "mfma", "kernel": "gemm_f8s_kernel / gemm_pp_kernel / gemm_f8_kernel / gemm_kernel<%s> (every encoder GEMM of rank 0's %d batches)" % (a.precision, nb),
                         "achieved": round(achieved, 2), "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_DENSE_TFLOPS, 4), "traffic": None,
                         "launches": gemm_n, "avg_launch_ms": round(gemm_ms / max(gemm_n, 1), 4), "gemm_ms_whole_set": round(gemm_ms, 1),
                         "note": "algorithmic FLOP = 2 M N K per launch; HIP events on the launch stream inside the timed region (rank 0)"},
            "seconds_whole_set": round(dt, 3), "host_synthesis_s": round(t_synth, 2),
            "end_to_end_frac_of_mfma_peak": round(value * gf / 1e3 / PEAK_BF16_DENSE_TFLOPS / R.world, 4) if gf else None,
            "hidden_checksum_per_rank": checks, "ranks": R.describe(rates), "build": source_hash()}
        # the sweep's full batches are the headline step's launches (same kernels, M = 96000): their measured traffic applies per launch (tail batch aside)
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and a.model == "small" and not a.trimmed and a.batch == 64:
            try:
                tj = json.load(open(tpath))
                hit = [e for e in tj.get("entries", [tj]) if e.get("build") == source_hash() and e.get("precision") == a.precision and e.get("weights", "fp16") == a.weights]
                if hit:
                    result["roofline"]["traffic"] = round(hit[-1]["per_kernel"]["gemm_kernel"]["hbm_bytes_per_launch"])
                    result["roofline"]["traffic_note"] = ("bytes per GEMM launch measured on the headline step of this build (profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE); "
                                                          "the sweep's %d full batches issue exactly those launches, its tail batch of %d clips smaller ones" % (nb - (1 if pcm.shape[0] % a.batch else 0), int(pcm.shape[0] % a.batch)))
            except Exception:
                pass
        if R.world == 1 and with_cpu_baseline:
            head = enc.encode_pcm(pcm[: a.batch])
            result["cpu_baseline"], result["parity"], _ = cpu_baseline_and_parity(a, torch, cfg, enc, pcm[: a.batch], shard[: a.batch], head)
    del enc, pcm
    torch.cuda.empty_cache()
    return result


def sweep_main(a):
    """BASELINE.json configs[4]: the whole seeded clip set, end to end, once.  Every rank pre-stages its contiguous shard as one
    int16 device tensor and walks it in batches of B (the last one short); value = clips of the whole set / max-over-ranks time."""
    from mlx8_ws_audio_transformer_amd import sweep
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    t0 = time.perf_counter()
    shard, first = sweep.stage_shard(a.clips, rank, world, seed=1234)     # host synthesis (forks a pool): before the GPU is touched
    t_synth = time.perf_counter() - t0
    R = Ranks(a)
    result = sweep_measure(a, R, shard, t_synth, not a.no_cpu_baseline)
    if R.rank == 0:
        result["wall_s"] = round(time.time() - T_START, 1)
        print(json.dumps(result))
    R.finish()


def device_pci_address(torch, dev):
    """'dddd:bb:dd.f' of the HIP device this rank runs on (sysfs names cards by it), or None when torch does not expose it."""
    try:
        p = torch.cuda.get_device_properties(dev)
        return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def read_power_sysfs(root: str = "/sys/class/drm", pci: str = None):
    """(package watts, shader clock MHz, which) from sysfs: hwmon power1_average (microwatts; power1_input on parts that only have that) and the
    starred line of pp_dpm_sclk -- of the card at PCI address `pci` (the device the benchmark runs on: `which` = "device <pci>") when sysfs shows
    it, else of the busiest amdgpu card of the node (`which` says so: on a shared multi-GPU host that can be another job's GPU).  None when no
    card exposes them.  Plain file reads: no rocm-smi, no child process, nothing is exec'ed."""
    import glob
    best = None
    devs = glob.glob(os.path.join(root, "card*", "device"))
    if pci:
        mine = [d for d in devs if os.path.basename(os.path.realpath(d)).lower() == pci.lower()]
        if mine:
            devs = mine
        else:
            pci = None
    for dev in devs:
        watts = None
        for name in ("power1_average", "power1_input"):
            for f in glob.glob(os.path.join(dev, "hwmon", "hwmon*", name)):
                try:
                    watts = int(open(f).read().strip()) / 1e6
                except (OSError, ValueError):
                    pass
            if watts is not None:
                break
        if watts is None:
            continue
        mhz = None
        try:
            for line in open(os.path.join(dev, "pp_dpm_sclk")):
                if "*" in line:
                    mhz = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
        except (OSError, ValueError, IndexError):
            pass
        if best is None or watts > best[0]:
            best = (watts, mhz, ("device " + pci) if pci else "busiest card of the node (the device's PCI address is not in sysfs)")
    return best


def profiler_attached() -> bool:
    """rocprofv3 / roctracer preloads are in the environment: no helper of any kind is started under a profiler."""
    return any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) or \
        any(k.startswith(("ROCPROFILER_", "ROCPROF_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


class PowerSampler:
    """Package power and shader clock while the headline workload loops, read from sysfs by a THREAD of this process every 0.2 s
    (round 2 ran rocm-smi -- a `#!/usr/bin/env python3` script -- from a helper process in a loop; under rocprofv3 that helper inherited
    a GPU-initialised image and every launch was a refused exec).  `window()` runs untimed steps and returns the median of the samples
    taken meanwhile; None when sysfs has no power file or a profiler is attached."""

    def __init__(self, root: str = "/sys/class/drm", period: float = 0.2, pci: str = None):
        import threading
        self.root, self.period, self.pci = root, period, pci
        self.rows, self._stop = [], threading.Event()
        self.thread = None
        if not profiler_attached() and read_power_sysfs(root, pci) is not None:
            self.thread = threading.Thread(target=self._run, daemon=True)
            self.thread.start()

    def _run(self):
        while not self._stop.wait(self.period):
            r = read_power_sysfs(self.root, self.pci)
            if r is not None:
                self.rows.append((time.time(), r[0], r[1], r[2]))

    def window(self, torch, step, dev, steps=60):
        if self.thread is None:
            return None
        torch.cuda.synchronize(dev)
        t0 = time.time()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        t1 = time.time()
        rows = [r for r in list(self.rows) if t0 + 0.3 <= r[0] <= t1]
        if not rows:
            return None
        med = lambda v: sorted(v)[len(v) // 2]
        clocks = [r[2] for r in rows if r[2] is not None]
        return {"package_w": round(med([r[1] for r in rows]), 1), "sclk_mhz": med(clocks) if clocks else None, "samples": len(rows), "card": rows[-1][3],
                "note": "median of sysfs samples (hwmon power1_average, pp_dpm_sclk) taken by a thread of this process while %d untimed steps of the headline workload ran" % steps}

    def close(self):
        self._stop.set()
        if self.thread is not None:
            self.thread.join(timeout=2)


def arithmetic_dtype(precision):
    """The operand type of the MFMA products (accumulation, residual stream, softmax and LayerNorm statistics are fp32 in every mode)."""
    return {"f16f8": "fp16", "fp16x3": "fp16", "fp16": "fp16", "bf16x3": "bf16", "bf16": "bf16"}.get(precision, "bf16")


def bench_weights(cfg, kind):
    """The deterministic random-init encoder weights of the bench (HF initialisation, seed 0); kind "fp16": every matrix rounded to fp16-representable
    values (fp32 tensors holding half-precision values, like a checkpoint converted from a half-precision release)."""
    import numpy as np
    from mlx8_ws_audio_transformer_amd import weights as wts
    W = wts.init_encoder_weights(cfg, 0, "hf")
    if kind == "fp16":
        W = {k: (v.astype(np.float16).astype(np.float32) if v.ndim >= 2 else v) for k, v in W.items()}
    return W


def load_bench_weights(torch, enc, W):
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    return enc


def cpu_budget():
    """(threads to use, note): the cores this process may actually RUN on = min(affinity mask, cgroup CPU quota).  On the GPU box the mask
    shows all 256 logical CPUs of the host while the container's share is one GPU's worth of it: the oracle on 256 threads ran 8x SLOWER
    than on 16 (0.152 vs 1.19 clips/s, 52 s vs 6.6 s per 8-clip pass; gpurun_out/r03/e3_bench.json) -- threads beyond the share only thrash."""
    avail = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                      # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                 # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    share = int(os.environ.get("AWT_CPU_THREADS", "0"))
    if share <= 0:
        share = max(1, int(quota)) if quota is not None else min(avail, 16)            # no readable quota: the documented share of a 1-GPU box
    n = max(1, min(avail, share))
    return n, "host shows %d logical CPUs; this process' CPU share is %s -> %d torch threads" % (
        avail, ("%.1f CPUs (cgroup quota)" % quota) if quota is not None else "taken as 16 (one GPU's share of the node; no cgroup quota readable; AWT_CPU_THREADS overrides)", n)


def cpu_baseline_and_parity(a, torch, cfg, enc, pcm, pcm_host, out_first):
    """The oracle (CPU restatement of the reference path) on the host cores this process owns (cpu_budget(): stated in `cores` /
    `threads_note`): 1 warm-up + up to 3 timed passes over a bounded sample, median, for mel alone, encoder alone and end
    to end; the timed passes stop once 45 s of CPU work are spent.  Then HIP vs oracle on the same clips."""
    import numpy as np
    from mlx8_ws_audio_transformer_amd import synth, weights as wts
    from oracle import encoder as oenc, logmel as omel
    n = min(a.cpu_clips, pcm.shape[0])
    threads, note = cpu_budget()
    torch.set_num_threads(threads)
    W = bench_weights(cfg, a.weights)
    clips_f32 = [synth.pcm_i16_to_f32(c) for c in pcm_host[:n]]
    med = lambda v: sorted(v)[len(v) // 2]
    t_mel, t_enc, spent = [], [], 0.0
    for it in range(4):                                     # pass 0 is the warm-up
        t0 = time.perf_counter()
        mel = omel.whisper_logmel(clips_f32, n_samples=cfg.n_frames * 160)
        t1 = time.perf_counter()
        with torch.no_grad():
            ref = oenc.encoder_forward(W, mel, cfg.heads)
        t2 = time.perf_counter()
        if it:
            t_mel.append(t1 - t0); t_enc.append(t2 - t1)
            spent += t2 - t0
            if spent > 45.0:
                break
    m_mel, m_enc, m_all = med(t_mel), med(t_enc), med([x + y for x, y in zip(t_mel, t_enc)])
    base = {"value": round(n / m_all, 3), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d of the step's clips, fp32, parity mode; 1 warm-up + %d timed passes, median (mel %.2f s + encoder %.2f s per pass)" % (n, len(t_enc), m_mel, m_enc),
            "threads_note": note, "mel_clips_per_s": round(n / m_mel, 3), "encoder_clips_per_s": round(n / m_enc, 3)}
    hid, feats = enc.encode_pcm(pcm[:n], return_features=True)
    e = oenc.error_norms(hid.cpu().numpy(), ref.numpy())
    parity = {"mel_max_abs": float(np.abs(feats.cpu().numpy() - mel).max()), "mel_tolerance": 1e-5,
              "hidden_max_abs": e["max_abs"], "hidden_mean_abs": e["mean_abs"], "hidden_rel_l2": e["rel_l2"],
              "hidden_tolerance": 1e-3, "norm_applied": "max_abs", "sample_clips": n}
    return base, parity, ref


def outlier_profile(a, torch, dev):
    """Hidden-state error of every parity mode vs the float64 oracle on the "realistic-outlier" weight profile
    (weights.with_outlier_channels: 30x LayerNorm gains, 10x fc2 / out_proj rows; Whisper-tiny, trimmed, 2 clips).  Product errors are
    relative, so absolute errors grow with the gains: this block shows which mode keeps the ABSOLUTE 1e-3 bound there."""
    import numpy as np
    from mlx8_ws_audio_transformer_amd import synth, weights as wts
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    from oracle import encoder as oenc, logmel as omel
    cfg = wts.config("tiny", True)
    W = wts.with_outlier_channels(wts.init_encoder_weights(cfg, 0, "test"), cfg, seed=0)
    clips = [synth.pcm_i16_to_f32(c) for c in synth.synth_clips_i16(2, seed=1234, first=0)]
    mel = omel.whisper_logmel(clips, n_samples=cfg.n_frames * 160)
    with torch.no_grad():
        ref64 = oenc.encoder_forward(W, mel, cfg.heads, dtype=torch.float64).numpy()
        ref32 = oenc.encoder_forward(W, mel, cfg.heads).numpy()
    res = {"weights": "Whisper-tiny (trimmed), 'test' init, 4 LayerNorm gains x30 per norm, 2 fc2 / out_proj rows x10", "ref_abs_max": float(np.abs(ref64).max()),
           "reference_fp32_vs_fp64_max_abs": float(np.abs(ref32 - ref64).max())}
    for prec in ["f16f8", "fp16x3", "bf16x3"]:
        enc = NativeWhisperEncoder(cfg, precision=prec, device=str(dev), seed=0, init_profile="test").eval()
        enc.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
        e = oenc.error_norms(enc(torch.from_numpy(mel).to(dev)).last_hidden_state.cpu().numpy(), ref64)
        res[prec] = {"max_abs": e["max_abs"], "mean_abs": e["mean_abs"], "rel_l2": e["rel_l2"]}
    return res


def gemm_algorithmic_bytes_per_step(exact_frac: float) -> float:
    """Operand + result bytes of one step's 50 GEMM launches (Whisper-small, B = 64): 3-byte operands for the projection matrices the library
    found fp16-exact, 4-byte operands for the rest (linear in the exact fraction; the conv stem is the same in both)."""
    return exact_frac * GEMM_ALGO_BYTES_PER_STEP_EXACT + (1.0 - exact_frac) * GEMM_ALGO_BYTES_PER_STEP


def measure_encode(a, R, _lib, enc, pcm, kind):
    """Warm-up, then EXACTLY a.steps timed passes of the hot path (barrier + synchronize on both sides, max over ranks) with the GEMM class
    event-timed live on the launch stream, then one extra untimed pass for the other classes' shares.  Returns the block both weight cases
    share: value, ms_per_step, roofline (with what the LIBRARY says about the weights, not what the command line asked for), time shares."""
    B, world = pcm.shape[0], R.world
    for _ in range(max(1, a.warmup)):
        enc.encode_pcm(pcm)
    R.sync()
    _lib.prof_enable(True, ["gemm"])
    for k in _lib.PROF_CLASSES:
        _lib.prof_collect(k)
    dt, mine, out = R.timed(lambda: enc.encode_pcm(pcm), a.steps)
    prof = {"gemm": _lib.prof_collect("gemm")}
    _lib.prof_enable(True, [k for k in _lib.PROF_CLASSES if k != "gemm"])
    enc.encode_pcm(pcm)
    for k in _lib.PROF_CLASSES:
        if k != "gemm":
            ms, n, fl = _lib.prof_collect(k)
            prof[k] = (ms * a.steps, n, fl)      # scaled so the per-step division below applies to every class
    _lib.prof_enable(False)
    gemm_ms, gemm_n, gemm_flop = prof["gemm"]
    achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    n_exact, n_mat = enc.exact16_matrices()
    exact_frac = n_exact / max(1, n_mat)
    terms = float(_lib.MFMA_PER_PAIR[enc.precision])
    if enc.precision == "f16f8":
        terms = 2.0 - 0.5 * exact_frac           # fp16 product + two e4m3 cross terms at twice the rate; an fp16-exact matrix drops one of them
    elif enc.precision == "fp16x3":
        terms = 3.0 - exact_frac
    block = {
        "value": round(B * world * a.steps / dt, 2), "unit": "clips/s", "ms_per_step": round(dt / a.steps * 1e3, 3),
        "weights": ("seed-0 random init, N(0, 0.02^2) linears as arbitrary fp32 values (SURVEY.md 8(d); what a checkpoint fine-tuned in fp32 holds)" if kind == "fp32"
                    else "the same initialisation rounded to fp16-representable values (a frozen base released in half precision)"),
        "fp16_exact_projection_matrices": "%d of %d (found by the library at upload)" % (n_exact, n_mat),
        "mfma_products_per_fragment_pair": round(terms, 3),
        "roofline": {"bound": "mfma", "kernel": "gemm_f8s_kernel / gemm_pp_kernel / gemm_f8_kernel / gemm_kernel<%s> (all encoder GEMMs: conv stem, QKV, out, fc1, fc2)" % enc.precision,
                     "achieved": round(achieved, 2), "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_BF16_DENSE_TFLOPS, 4), "traffic": None,
                     "launches": gemm_n, "avg_launch_ms": round(gemm_ms / max(gemm_n, 1), 4),
                     "mfma_issue_frac": round(terms * achieved / PEAK_BF16_DENSE_TFLOPS, 4)},
        "time_share_ms_per_step": {k: round(v[0] / a.steps, 3) for k, v in prof.items()},
    }
    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (tools/profile_round.sh); a committed summary is
    # attached only when it was measured on THIS build (source hash), this precision and this kind of weights
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and a.model == "small" and not a.trimmed and B == 64:
        try:
            tj = json.load(open(tpath))
            entries = tj.get("entries", [tj])
            hit = [e for e in entries if e.get("build") == source_hash() and e.get("precision") == enc.precision and e.get("weights", "fp16") == kind]
            if hit:
                block["roofline"]["traffic"] = round(hit[-1]["per_kernel"]["gemm_kernel"]["hbm_bytes_per_launch"])
                block["roofline"]["traffic_note"] = "bytes per GEMM launch, FETCH_SIZE x2 (gfx950) + WRITE_SIZE, rocprofv3 --pmc passes on this build (profiles/traffic.json)"
                block["roofline"]["algorithmic_bytes_per_launch"] = round(gemm_algorithmic_bytes_per_step(exact_frac) / 50)
            else:
                block["roofline"]["traffic_note"] = "profiles/traffic.json has no entry for build %s / %s / %s weights: not attached" % (source_hash(), enc.precision, kind)
        except Exception:
            pass
    gf = ENCODER_GFLOP_PER_CLIP.get((a.model, a.trimmed))
    if gf:
        block["end_to_end_algorithmic_tflops"] = round(block["value"] * gf / 1e3, 2)
        block["end_to_end_frac_of_mfma_peak"] = round(block["value"] * gf / 1e3 / PEAK_BF16_DENSE_TFLOPS / world, 4)
    return block, dt, mine, out


def other_configs(a, R, torch, dev, _lib, staged):
    """Compact blocks of BASELINE.json's other 1-GPU configurations, produced by the same code paths as their own workloads
    (--workload finetune / sweep; the tiny encoder through measure_encode): configs[1] Whisper-tiny B = 32, configs[2] the fine-tune step,
    configs[4]'s 1-GPU leg.  Each carries value, ms_per_step, roofline.frac and a parity figure."""
    import copy
    import numpy as np
    from mlx8_ws_audio_transformer_amd import synth, weights as wts
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder
    from oracle import encoder as oenc, logmel as omel
    out = {}
    # ---- configs[1]: Whisper-tiny encoder forward, batch = 32 x 4 s clips
    at = copy.copy(a); at.model, at.batch, at.trimmed = "tiny", 32, False
    cfg = wts.config("tiny", False)
    pcm_host = synth.synth_clips_i16(32, seed=1234, first=0)
    pcm = torch.from_numpy(pcm_host).to(dev)
    enc = load_bench_weights(torch, NativeWhisperEncoder(cfg, precision=a.precision, device=str(dev), seed=0, init_profile="hf").eval(), bench_weights(cfg, a.weights))
    blk, _, _, hid = measure_encode(at, R, _lib, enc, pcm, a.weights)
    mel = omel.whisper_logmel([synth.pcm_i16_to_f32(c) for c in pcm_host[:2]], n_samples=cfg.n_frames * 160)
    with torch.no_grad():
        ref = oenc.encoder_forward(bench_weights(cfg, a.weights), mel, cfg.heads).numpy()
    out["tiny_b32"] = {"config": "BASELINE.json configs[1]: Whisper-tiny encoder forward (log-mel included), batch 32 x 4 s clips, %s, parity mode" % a.precision,
                       "value": blk["value"], "unit": "clips/s", "ms_per_step": blk["ms_per_step"], "roofline": blk["roofline"],
                       "time_share_ms_per_step": blk["time_share_ms_per_step"],
                       "parity": {"hidden_max_abs": float(np.abs(hid[:2].cpu().numpy() - ref).max()), "hidden_tolerance": 1e-3, "sample_clips": 2}}
    del enc, pcm
    # ---- configs[2]: Whisper-small + LoRA r = 8 fine-tune step, batch 64 (library default backward, and the f16f8-MLP form beside it)
    af = copy.copy(a); af.model, af.batch, af.trimmed, af.precision, af.steps, af.warmup = "small", 64, False, "bf16x3", min(a.steps, 3), 1
    af.lora_r, af.torch_decoder, af.decoder_dtype = 8, False, "fp32"
    for key, bwd in (("finetune_step", None), ("finetune_step_f16f8_mlp", "f16f8")):
        r = finetune_measure(af, R, bwd, False)
        out[key] = {"config": "BASELINE.json configs[2]: Whisper-small + LoRA r=8 (q_proj, v_proj) fine-tune step, batch 64, %s" % (
                        "every GEMM split-bf16 (library default)" if bwd is None else "the MLP's four GEMMs per layer in f16f8"),
                    "value": r["value"], "unit": "clips/s", "ms_per_step": r["ms_per_step"], "roofline": r["roofline"],
                    "roofline_attention_backward": r["roofline_attention_backward"], "time_share_ms_per_step": r["time_share_ms_per_step"], "last_loss": r["last_loss"],
                    "parity": "gradients vs torch autograd on the fp32 oracle: tests/test_gpu_backward.py (worst relative error 2.9e-5 default, <= 2e-3 f16f8 MLP)"}
    # ---- configs[4], 1-GPU leg: the 10k-clip sweep (shard staged before the GPU was touched)
    if staged is not None:
        asw = copy.copy(a); asw.model, asw.batch, asw.trimmed, asw.clips, asw.warmup = "small", 64, False, staged[2], 1
        r = sweep_measure(asw, R, staged[0], staged[1], False)
        out["sweep_10k"] = {"config": "BASELINE.json configs[4], 1-GPU leg: %d-clip seeded set, batches of 64 incl. the tail batch, log-mel + Whisper-small encoder" % staged[2],
                            "value": r["value"], "unit": "clips/s", "ms_per_step": r["ms_per_step"], "seconds_whole_set": r["seconds_whole_set"], "roofline": r["roofline"],
                            "hidden_checksum": r["hidden_checksum_per_rank"], "host_synthesis_s": r["host_synthesis_s"]}
    return out


def encode_main(a):
    sampler = PowerSampler() if (int(os.environ.get("RANK", "0")) == 0 and a.gpus == 1 and not a.no_power) else None   # a sysfs-reading thread; never under a profiler
    staged = None
    if a.gpus == 1 and "WORLD_SIZE" not in os.environ and not a.no_configs and a.model == "small" and not a.trimmed:
        from mlx8_ws_audio_transformer_amd import sweep as _sweep
        t0 = time.perf_counter()
        shard, _ = _sweep.stage_shard(a.clips, 0, 1, seed=1234)        # the sweep block's clip set: host synthesis forks a pool, so before the GPU is touched
        staged = (shard, time.perf_counter() - t0, a.clips)
    R = Ranks(a)
    torch, dev, rank, world = R.torch, R.dev, R.rank, R.world
    if sampler is not None:
        sampler.pci = device_pci_address(torch, dev)      # from here on the samples come from the card this rank runs on, not the node's busiest
    from mlx8_ws_audio_transformer_amd import _lib, synth, weights as wts
    from mlx8_ws_audio_transformer_amd.encoder import NativeWhisperEncoder

    cfg = wts.config(a.model, a.trimmed)
    B = a.batch
    # synthetic clips: this rank's contiguous shard of the seeded piano-note set
    pcm_host = synth.synth_clips_i16(B, seed=1234, first=rank * B)
    pcm = torch.from_numpy(pcm_host).to(dev)

    def build(kind, precision=a.precision):
        return load_bench_weights(torch, NativeWhisperEncoder(cfg, precision=precision, device=str(dev), chunk_clips=a.chunk, seed=0, init_profile="hf").eval(),
                                  bench_weights(cfg, kind))

    enc = build(a.weights)
    head, dt, mine, out = measure_encode(a, R, _lib, enc, pcm, a.weights)
    rates = R.gather(B * a.steps / mine)
    result = {
        "metric": "4s@16kHz clips/sec through mel+Whisper-%s encoder" % a.model,
        "value": head["value"], "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": arithmetic_dtype(a.precision), "data": "synthetic", "device": device_info(torch, dev),
        "config": {"workload": "int16 PCM [B,64000] in HBM -> Whisper log-mel [B,80,%d] -> Whisper-%s encoder -> hidden [B,%d,%d] fp32"
                               % (cfg.n_frames, a.model, cfg.max_source_positions, cfg.d_model),
                   "mode": "trimmed (NOT reference-equivalent)" if a.trimmed else "parity (clip zero-padded to 30 s, as the reference computes)",
                   "clips_per_gpu_per_step": B, "precision": a.precision,
                   "mfma_products_per_fragment_pair": head["mfma_products_per_fragment_pair"], "weights": head["weights"],
                   "fp16_exact_projection_matrices": head["fp16_exact_projection_matrices"],
                   "parallelism": "dp%d (clip shards, no data-path collective)" % world},
        "roofline": head["roofline"], "time_share_ms_per_step": head["time_share_ms_per_step"],
        "ranks": R.describe(rates), "build": source_hash(),
    }
    for k in ("end_to_end_algorithmic_tflops", "end_to_end_frac_of_mfma_peak"):
        if k in head:
            result[k] = head[k]

    side_out = {}
    if rank == 0 and world == 1:
        # ---- package power and shader clock while the headline workload runs (sysfs samples on a thread of this process, outside the timed
        #      region): the encoder sits at the package power limit, which is what caps roofline.frac
        result["power"] = sampler.window(torch, lambda: enc.encode_pcm(pcm), dev) if sampler else None
        if not a.no_fast_mode:
            # ---- the other kind of weights, same build and mode, as a first-class block with its own roofline
            if a.precision in ("f16f8", "fp16x3"):
                okind = "fp16" if a.weights == "fp32" else "fp32"
                other = build(okind)
                oblock, _, _, _ = measure_encode(a, R, _lib, other, pcm, okind)
                result["fp16_exact_weights" if okind == "fp16" else "general_fp32_weights"] = oblock
                del other
            # ---- the other operand modes, reported beside the headline with all three error norms vs the headline's output
            side = {}
            for prec in PRECISIONS:
                if prec == a.precision:
                    continue
                other = build(a.weights, prec)
                for _ in range(max(1, a.warmup)):
                    other.encode_pcm(pcm)
                fdt, _, fout = R.timed(lambda: other.encode_pcm(pcm), a.steps)
                d = (fout[: a.cpu_clips].double() - out[: a.cpu_clips].double())
                side[prec] = {"value": round(B * a.steps / fdt, 2), "unit": "clips/s", "ms_per_step": round(fdt / a.steps * 1e3, 3),
                              "vs_headline_max_abs": float(d.abs().max()), "vs_headline_mean_abs": float(d.abs().mean()),
                              "vs_headline_rel_l2": float(d.norm() / out[: a.cpu_clips].double().norm()),
                              "mfma_products_per_fragment_pair": _lib.MFMA_PER_PAIR[prec]}
                side_out[prec] = fout[: a.cpu_clips].cpu().numpy()
                del other
            result["other_precisions"] = side
        # ---- CPU baseline: the oracle (CPU restatement of the reference path) on this host's cores, bounded sample
        if not a.no_cpu_baseline:
            result["cpu_baseline"], result["parity"], ref = cpu_baseline_and_parity(a, torch, cfg, enc, pcm, pcm_host, out)
            # every side mode against the SAME oracle output, all three norms of SURVEY.md 0.5, and which of them meets 1e-3 (the headline applies max-abs)
            from oracle import encoder as _oenc
            for prec, arr in side_out.items():
                e = _oenc.error_norms(arr[: ref.shape[0]], ref.numpy())
                result["other_precisions"][prec]["vs_oracle"] = {k: float(e[k]) for k in ("max_abs", "mean_abs", "rel_l2")}
                result["other_precisions"][prec]["meets_1e-3"] = {k: bool(e[k] <= 1e-3) for k in ("max_abs", "mean_abs", "rel_l2")}
            result["outlier_profile"] = outlier_profile(a, torch, dev)
        if not a.no_configs and a.model == "small" and not a.trimmed:
            del enc
            torch.cuda.empty_cache()
            result["configs"] = other_configs(a, R, torch, dev, _lib, staged)
    if sampler:
        sampler.close()
    if rank == 0:
        result["wall_s"] = round(time.time() - T_START, 1)
        print(json.dumps(result))
    R.finish()


def main():
    a = parse()
    if a.precision is None:
        a.precision = "bf16x3" if a.workload == "finetune" else "f16f8"
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a))                 # the parent never initialises the GPU
    {"encode": encode_main, "sweep": sweep_main, "finetune": finetune_main, "noop": noop_main}[a.workload](a)


if __name__ == "__main__":
    main()
