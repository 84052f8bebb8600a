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


# ---------------------------------------------------------------- the real Seq2SeqTrainer.training_step on two gloo ranks
class _CpuAdapterModel(torch.nn.Module):
    """CPU stand-in with the model surface the trainer uses (`lora_parameters()`, `.encoder.device`, forward -> `.loss`):
    frozen linear + LoRA pair.  The native encoder needs a GPU; the trainer's own logic (micro-batches, flat in-place
    exchange, clipping on the flat buffer, AdamW, schedule) does not."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.w = torch.nn.Parameter(torch.randn(16, 32, generator=g), requires_grad=False)
        self.A = torch.nn.Parameter(torch.randn(4, 32, generator=g) * 0.3)
        self.B = torch.nn.Parameter(torch.randn(16, 4, generator=g) * 0.3)
        from types import SimpleNamespace
        self.encoder = SimpleNamespace(device=torch.device("cpu"))

    def lora_parameters(self):
        return [self.A, self.B]

    def forward(self, input_features, labels):
        from types import SimpleNamespace
        y = input_features @ self.w.t() + (input_features @ self.A.t()) @ self.B.t()
        return SimpleNamespace(loss=torch.nn.functional.cross_entropy(y, labels))


def _trainer_batch():
    g = torch.Generator().manual_seed(11)
    return {"input_features": torch.randn(8, 32, generator=g), "labels": torch.randint(0, 16, (8,), generator=g)}


def _train_two_steps(batches):
    from mlx8_ws_audio_transformer_amd.finetune import Seq2SeqTrainer, Seq2SeqTrainingArguments
    model = _CpuAdapterModel()
    args = Seq2SeqTrainingArguments(learning_rate=1e-2, warmup_steps=0, max_steps=4, max_grad_norm=0.5, predict_with_generate=False)
    tr = Seq2SeqTrainer(args=args, model=model)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(tr.bucket.params, tr.bucket.views))   # .grad ARE the flat buffer
    losses = [tr.training_step(b) for b in batches]
    return losses, torch.cat([model.A.detach().flatten(), model.B.detach().flatten()]), tr


def _trainer_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = _trainer_batch()
    lo, hi = shard_range(8, rank, world)
    mine = {k: v[lo:hi] for k, v in full.items()}
    # step 1: one micro-batch; step 2: gradient accumulation over two micro-batches of the shard
    losses, params, tr = _train_two_steps([mine, [{k: v[:2] for k, v in mine.items()}, {k: v[2:] for k, v in mine.items()}]])
    assert "gloo" in tr.exchange
    ret[rank] = (losses, params)
    dist.destroy_process_group()


def test_training_step_on_two_ranks_equals_the_full_batch_step():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, ret)) for r in range(world)]
        [p.start() for p in procs]
        [p.join(180) for p in procs]
        assert all(p.exitcode == 0 for p in procs)
        (l0, p0), (l1, p1) = ret[0], ret[1]
    assert torch.equal(p0, p1)                                   # every rank took the same optimizer steps
    full = _trainer_batch()
    # single process, whole batch: equal shards and equal micro-batches, so mean of means == global mean
    _, want, tr = _train_two_steps([full, full])
    assert tr.exchange == "none"
    torch.testing.assert_close(p0, want, rtol=2e-5, atol=2e-6)
    with torch.no_grad():
        first = float(_CpuAdapterModel()(full["input_features"], full["labels"]).loss)
    assert abs((l0[0] + l1[0]) / 2 - first) < 1e-5


def test_bucket_rebinds_foreign_gradients_and_clips_like_torch():
    a = torch.nn.Parameter(torch.ones(3)); b = torch.nn.Parameter(torch.ones(2, 2))
    bk = FlatGradBucket([a, b])
    (a.sum() * 3 + (b * b).sum()).backward()                      # autograd accumulates INTO the views
    assert torch.equal(bk.flat, torch.tensor([3., 3., 3., 2., 2., 2., 2.]))
    a.grad = None; b.grad = torch.full((2, 2), 5.0)               # gradients replaced behind the bucket's back
    assert bk.bind(keep=True) == 2 and torch.equal(bk.flat, torch.tensor([0., 0., 0., 5., 5., 5., 5.]))
    ref = [torch.nn.Parameter(torch.zeros(7))]; ref[0].grad = bk.flat.clone()
    want = torch.nn.utils.clip_grad_norm_(ref, 1.0)
    got = bk.clip_norm_(1.0)
    torch.testing.assert_close(got, want)
    torch.testing.assert_close(bk.flat, ref[0].grad)
