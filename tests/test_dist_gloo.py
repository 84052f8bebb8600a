"""N > 1 path on CPU: world_size-2 gloo process group, clip sharding and the single flat all-reduce of adapter gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mlx8_ws_audio_transformer_amd.dist import FlatGradBucket, shard_range


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 64, 10000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            covered = [i for lo, hi in spans for i in range(lo, hi)]
            assert covered == list(range(n))
    assert shard_range(10000, 3, 8) == (3750, 5000)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    # the same "adapters" on every rank, per-rank data shard -> per-rank gradients
    A = torch.nn.Parameter(torch.randn(8, 32)); Bm = torch.nn.Parameter(torch.randn(32, 8) * 0.1)
    frozen = torch.nn.Parameter(torch.randn(4), requires_grad=False)
    data = torch.arange(6 * 32, dtype=torch.float32).reshape(6, 32) / 100
    lo, hi = shard_range(6, rank, world)
    x = data[lo:hi]
    loss = ((x @ A.t()) @ Bm.t()).square().mean()
    loss.backward()
    bucket = FlatGradBucket([A, Bm, frozen])
    assert bucket.numel == A.numel() + Bm.numel()
    bucket.allreduce_mean()
    ret[rank] = torch.cat([A.grad.flatten(), Bm.grad.flatten()])
    dist.destroy_process_group()


def test_flat_allreduce_equals_full_batch_gradient():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
        [p.start() for p in procs]
        [p.join(120) for p in procs]
        assert all(p.exitcode == 0 for p in procs)
        g0, g1 = ret[0], ret[1]
    assert torch.equal(g0, g1)
    # equal shards: mean of per-shard mean-loss gradients == gradient of the full-batch mean loss
    torch.manual_seed(0)
    A = torch.nn.Parameter(torch.randn(8, 32)); Bm = torch.nn.Parameter(torch.randn(32, 8) * 0.1)
    data = torch.arange(6 * 32, dtype=torch.float32).reshape(6, 32) / 100
    ((data @ A.t()) @ Bm.t()).square().mean().backward()
    full = torch.cat([A.grad.flatten(), Bm.grad.flatten()])
    torch.testing.assert_close(g0, full, rtol=1e-5, atol=1e-6)


def test_bucket_is_noop_without_process_group():
    p = torch.nn.Parameter(torch.ones(3)); p.grad = torch.full((3,), 2.0)
    b = FlatGradBucket([p]); b.allreduce_mean()
    assert torch.equal(p.grad, torch.full((3,), 2.0))
