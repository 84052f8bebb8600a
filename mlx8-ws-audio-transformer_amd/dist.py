"""Data-parallel plumbing for the LoRA fine-tune step (BASELINE.json configs[3]).

The reference has no distributed code; this is build-defined and deliberately small:
one process per GPU, clips sharded contiguously across ranks, and ONE flat fp32 buffer holding every adapter
gradient (r = 16 on q, v of Whisper-small: 589 824 floats = 2.36 MB) that is averaged over the ranks IN PLACE:

* the adapter parameters' `.grad` tensors are views into that buffer and the native backward writes them there
  (`encoder.NativeWhisperEncoder.bind_grad_buffer`), so nothing is packed or unpacked;
* on the GPU the exchange is RCCL over xGMI through libawt's own communicator (`awt_comm_*`, `awt_allreduce_mean_f32`;
  include/awt.h), issued from inside `awt_encoder_backward_ex` on a side stream layer group by layer group, so the
  upper layers' gradients travel while the lower layers' backward still runs;
* `torch.distributed` is kept for rendezvous (the 128-byte RCCL id travels through its store), for barriers / timing,
  and as the exchange itself under the "gloo" backend of the CPU tests.

Inference / throughput sweeps need no collective at all.
"""
from __future__ import annotations

import ctypes as C
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


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class AwtComm:
    """libawt's RCCL communicator over the ranks of the default torch.distributed group (the 128-byte id is broadcast
    through torch.distributed; every rank must construct it).  `world=1` without a process group gives a single-rank
    communicator (the all-reduce is then RCCL's in-place identity) -- used to exercise the native path on one GPU."""

    def __init__(self, device: torch.device):
        from . import _lib
        L = _lib.lib()
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = world_size()
        ident = (C.c_char * _lib.COMM_ID_BYTES)()
        if rank == 0:
            _lib.check(L.awt_comm_unique_id(C.cast(ident, C.c_void_p)))
        if world > 1:
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0)
            ident = (C.c_char * _lib.COMM_ID_BYTES).from_buffer_copy(box[0])
        out = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(L.awt_comm_create(_lib.ctx(device), C.cast(ident, C.c_void_p), rank, world, C.byref(out)))
        self.handle, self.world, self.rank, self.device = out.value, world, rank, device

    def allreduce_mean_(self, flat: torch.Tensor) -> None:
        from . import _lib
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().awt_allreduce_mean_f32(self.handle, _lib.ptr(flat), flat.numel(), _lib.stream_handle()))

    def allreduce_sum_(self, flat: torch.Tensor) -> None:
        from . import _lib
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().awt_allreduce_sum_f32(self.handle, _lib.ptr(flat), flat.numel(), _lib.stream_handle()))

    def bucket_stats(self, enable: bool = True):
        """[(milliseconds on the side stream, bytes)] of the bucket reductions the native backward issued since the last call (include/awt.h
        awt_comm_bucket_stats); `enable` keeps / stops the recording."""
        from . import _lib
        ms = (C.c_double * 8)(); nb = (C.c_int64 * 8)(); n = C.c_int(0)
        _lib.check(_lib.lib().awt_comm_bucket_stats(self.handle, 1 if enable else 0, ms, nb, 8, C.byref(n)))
        return [(float(ms[i]), int(nb[i])) for i in range(n.value)]

    def close(self) -> None:
        if self.handle is not None:
            from . import _lib
            _lib.lib().awt_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FlatGradBucket:
    """One contiguous fp32 buffer whose slices ARE the parameters' gradients: `p.grad` of every trainable parameter is
    bound to a view of `flat`, so autograd accumulates into it in place and the exchange needs no pack / unpack copies.
    `flat` may be supplied (e.g. with a leading segment that libawt's backward writes directly).  A parameter whose
    `.grad` was replaced behind the bucket's back (optimizer.zero_grad(set_to_none=True), a fresh autograd tensor) is
    copied in and re-bound before the exchange."""

    def __init__(self, params: Iterable[torch.nn.Parameter], flat: Optional[torch.Tensor] = None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        if flat is None:
            flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        if flat.numel() != self.numel or flat.dtype != torch.float32 or not flat.is_contiguous():
            raise ValueError("flat must be a contiguous float32 tensor with one element per trainable parameter element")
        self.flat = flat
        self.views: List[torch.Tensor] = []
        off = 0
        for p in self.params:
            n = p.numel()
            self.views.append(flat[off: off + n].view_as(p))
            off += n
        self.bind(keep=True)

    def bind(self, keep: bool = False) -> int:
        """Make every p.grad the bucket's view (copying a foreign gradient in when `keep`); returns how many were re-bound."""
        n = 0
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is not None and g.data_ptr() == v.data_ptr() and g.shape == v.shape:
                continue
            if g is not None and keep:
                v.copy_(g)
            elif g is None and keep:
                v.zero_()
            p.grad = v
            n += 1
        return n

    def zero(self) -> None:
        self.flat.zero_()

    def allreduce_mean(self, world: Optional[int] = None, comm: Optional[AwtComm] = None, segment: Optional[slice] = None) -> None:
        """grad <- mean over ranks, in place (`segment`: only that slice of the flat buffer); a no-op outside a process group.
        `comm`: libawt's RCCL communicator (GPU); otherwise torch.distributed's all_reduce (gloo in the CPU tests)."""
        self.bind(keep=True)
        buf = self.flat if segment is None else self.flat[segment]
        if buf.numel() == 0:
            return
        if comm is not None:
            comm.allreduce_mean_(buf)
            return
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = world or dist.get_world_size()
        if world == 1:
            return
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        buf.div_(world)

    def clip_norm_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ on the flat buffer (two kernels instead of a few per parameter); returns the norm."""
        total = self.flat.norm(2)
        self.flat.mul_((max_norm / (total + 1e-6)).clamp(max=1.0))
        return total
