"""GPU: the data-parallel fine-tune step (BASELINE.json configs[3]) as far as ONE GPU can show it:
* two ranks (gloo rendezvous, both on cuda:0) run the real `Seq2SeqTrainer.training_step` on their clip shards and end on
  the parameters of the single-process full-batch step -- the gradients are exchanged in place in the flat buffer;
* libawt's own RCCL communicator (`awt_comm_*`, `awt_allreduce_*_f32`, the in-backward side-stream exchange of
  `awt_encoder_backward_ex`) runs for real with one rank, where averaging is the identity.
RCCL refuses two ranks on one device, so the N > 1 RCCL exchange itself is exercised by the driver's 8-GPU run only."""
import os
import socket

import numpy as np
import pytest
import torch

from mlx8_ws_audio_transformer_amd import weights as wts
from oracle import logmel as oracle_mel
from tests.util import piano_clips_f32

pytestmark = pytest.mark.gpu


def _batch(cfg, B):
    mel = oracle_mel.whisper_logmel(piano_clips_f32(B, 40), n_samples=cfg.n_frames * 160)
    g = torch.Generator().manual_seed(9)
    labels = torch.randint(3, 1000, (B, 6), generator=g)
    labels[:, 0] = 50258
    return {"input_features": torch.from_numpy(mel), "labels": labels}


def _trainer(cfg, lr=1e-2):
    from mlx8_ws_audio_transformer_amd.finetune import Seq2SeqTrainer, Seq2SeqTrainingArguments, WhisperLoRAModel
    model = WhisperLoRAModel(cfg, wts.LoraSpec(r=8, alpha=16.0), decoder_layers=1)
    with torch.no_grad():
        for p in model.lora_parameters():
            if p.shape[1] == 8:
                p.copy_(torch.from_numpy(0.05 * wts.unit_variates("dp", p.numel(), 1).reshape(p.shape).astype(np.float32)))
    args = Seq2SeqTrainingArguments(learning_rate=lr, warmup_steps=0, max_steps=8, max_grad_norm=1.0, predict_with_generate=False)
    return model, Seq2SeqTrainer(args=args, model=model)


def _adapters(model):
    return torch.cat([p.detach().flatten() for p in model.encoder.lora_parameters_library_order()]).cpu()


def _rank_main(rank, world, port, path):
    import torch.distributed as dist
    from mlx8_ws_audio_transformer_amd.dist import shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = wts.config("mini", True)
    full = _batch(cfg, 4)
    lo, hi = shard_range(4, rank, world)
    mine = {k: v[lo:hi] for k, v in full.items()}
    model, tr = _trainer(cfg)
    assert "gloo" in tr.exchange and tr.comm is None
    losses = [tr.training_step(mine), tr.training_step([{k: v[:1] for k, v in mine.items()}, {k: v[1:] for k, v in mine.items()}])]
    torch.save({"losses": losses, "adapters": _adapters(model)}, os.path.join(path, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_gloo_ranks_take_the_full_batch_step(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    [p.start() for p in procs]
    [p.join(600) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    assert torch.equal(r0["adapters"], r1["adapters"])
    cfg = wts.config("mini", True)
    full = _batch(cfg, 4)
    model, tr = _trainer(cfg)
    assert tr.exchange == "none"
    tr.training_step(full); tr.training_step(full)
    want = _adapters(model)
    start = _adapters(_trainer(cfg)[0])
    moved = float((want - start).abs().max())
    assert moved > 1e-3
    # AdamW divides by |g|: elements whose gradient is near zero amplify the (1e-3 relative) shard-vs-full-batch rounding noise
    assert float((r0["adapters"] - want).abs().max()) < 1e-2 * moved


def test_libawt_rccl_communicator_single_rank():
    from mlx8_ws_audio_transformer_amd.dist import AwtComm
    comm = AwtComm(torch.device("cuda", 0))
    assert comm.world == 1 and comm.handle
    x = torch.from_numpy(wts.unit_variates("ar", 1 << 20, 0).astype(np.float32)).cuda()
    y = x.clone()
    comm.allreduce_sum_(y); comm.allreduce_mean_(y)
    torch.cuda.synchronize()
    assert torch.equal(x, y)                      # one rank: sum and mean are the identity, computed by RCCL in place
    comm.close()


def test_in_backward_exchange_matches_the_plain_step():
    """AWT_BWD_ALLREDUCE on a one-rank communicator: the side-stream exchange of both layer groups and the accumulate path
    run, and the step equals the one taken without a communicator."""
    cfg = wts.config("mini", True)
    full = _batch(cfg, 4)
    micro = [{k: v[:2] for k, v in full.items()}, {k: v[2:] for k, v in full.items()}]
    out = {}
    for native in (False, True):
        model, tr = _trainer(cfg)
        if native:
            tr._setup_exchange(force_native=True)
            assert tr.comm is not None and "rccl" in tr.exchange
        losses = [tr.training_step(full), tr.training_step(micro)]
        torch.cuda.synchronize()
        out[native] = (losses, _adapters(model))
    assert out[True][0] == out[False][0]
    assert torch.equal(out[True][1], out[False][1])
