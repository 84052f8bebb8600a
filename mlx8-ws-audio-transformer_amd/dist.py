"""Data-parallel plumbing for the LoRA fine-tune step (BASELINE.json configs[3]).

The reference has no distributed code; this is build-defined and deliberately small:
one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests),
clips sharded contiguously across ranks, and ONE all-reduce per step over a single flat fp32 buffer holding every
adapter gradient (r = 16 on q, v of Whisper-small: 589 824 floats = 2.36 MB -- latency-bound, so never bucketed).
Inference / throughput sweeps need no collective at all.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from RANK / LOCAL_RANK / WORLD_SIZE; initialises the process group when world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Rank r of N takes clips [r * ceil(M / N), (r + 1) * ceil(M / N)) clipped to M."""
    per = -(-n_items // world)
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


class FlatGradBucket:
    """One contiguous fp32 buffer aliasing nothing: gradients are copied in, reduced with a single collective, averaged
    and copied back.  Keeping the buffer between steps avoids per-step allocation."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)

    def pack(self) -> torch.Tensor:
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off: off + n].zero_()
            else:
                self.flat[off: off + n].copy_(p.grad.reshape(-1))
            off += n
        return self.flat

    def unpack(self) -> None:
        off = 0
        for p in self.params:
            n = p.numel()
            g = self.flat[off: off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n

    def allreduce_mean(self, world: Optional[int] = None) -> None:
        """grad <- mean over ranks; a no-op outside a process group."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = world or dist.get_world_size()
        if world == 1:
            return
        self.pack()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(world)
        self.unpack()
