"""The UrbanSound Transformer classifier's inference path on the native kernels (d 128, heads of 32, post-LN, S = T + 1)
against the torch.nn restatement of the reference module (oracle/urbansound_classifier.py)."""
import numpy as np
import pytest
import torch

from mlx8_ws_audio_transformer_amd import weights as wts
from oracle.urbansound_classifier import ReferenceTransformerClassifier

pytestmark = pytest.mark.gpu


def _pair(n_mels, T, seed):
    from mlx8_ws_audio_transformer_amd.urbansound_classifier import TransformerUrbanSound8KClassifier
    torch.manual_seed(seed)
    ref = ReferenceTransformerClassifier(n_mels=n_mels).eval()
    with torch.no_grad():                       # make every parameter non-trivial (fresh LayerNorms / biases are 1 / 0)
        for i, (name, p) in enumerate(ref.named_parameters()):
            u = torch.from_numpy(wts.unit_variates("cls." + name, p.numel(), seed).reshape(p.shape).astype(np.float32))
            p.copy_(1.0 + 0.1 * u if "norm" in name and name.endswith("weight") else (p + 0.05 * u if p.dim() > 1 else 0.1 * u))
        ref.pos_embed = torch.nn.Parameter(0.02 * torch.randn(1, T + 1, 128)); ref.n_frames = T
    nat = TransformerUrbanSound8KClassifier(n_mels=n_mels)
    nat.pos_embed = torch.nn.Parameter(ref.pos_embed.detach().clone()); nat.n_frames = T
    missing, unexpected = nat.load_state_dict(ref.state_dict(), strict=True)      # same names as the reference module
    assert not missing and not unexpected
    return ref, nat.cuda().eval()


@pytest.mark.parametrize("n_mels,T,batch", [(128, 126, 3), (80, 501, 2), (64, 126, 5)])
def test_logits_and_features_match_reference_module(n_mels, T, batch):
    ref, nat = _pair(n_mels, T, seed=n_mels + T)
    x = torch.from_numpy(wts.unit_variates("cls.x", batch * n_mels * T, 3).reshape(batch, n_mels, T).astype(np.float32)) * 2.0 - 4.0
    with torch.no_grad():
        want_f, want = ref.features(x), ref(x)
    got_f, got = nat.get_feature_embeddings(x.cuda()).cpu(), nat(x.cuda()).cpu()
    assert tuple(got.shape) == (batch, 10) and tuple(got_f.shape) == (batch, 128)
    np.testing.assert_allclose(got_f.numpy(), want_f.numpy(), rtol=0, atol=1e-3)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-3)
    assert torch.equal(got.argmax(-1), want.argmax(-1))


def _train_pair(n_mels, T, seed, dropout=0.0):
    from mlx8_ws_audio_transformer_amd.urbansound_classifier import TransformerUrbanSound8KClassifier
    ref, _ = _pair(n_mels, T, seed)
    ref0 = ReferenceTransformerClassifier(n_mels=n_mels, dropout=dropout)
    ref0.pos_embed = torch.nn.Parameter(ref.pos_embed.detach().clone()); ref0.n_frames = T
    ref0.load_state_dict(ref.state_dict())
    nat = TransformerUrbanSound8KClassifier(n_mels=n_mels, dropout=dropout)
    nat.pos_embed = torch.nn.Parameter(ref.pos_embed.detach().clone()); nat.n_frames = T
    nat.load_state_dict(ref.state_dict())
    return ref0.train(), nat.cuda().train()


@pytest.mark.parametrize("n_mels,T,batch", [(128, 126, 4), (64, 501, 2)])
def test_training_step_gradients_match_autograd_of_the_reference_module(n_mels, T, batch):
    """One forward + backward in train() mode with dropout 0: loss and the gradient of EVERY parameter against torch autograd over
    the reference module's restatement (fp32 CPU).  Tolerance: 2e-4 of each gradient's largest magnitude (bf16x3 GEMMs, the
    weight-gradient contraction runs over B (T + 1) rows)."""
    from mlx8_ws_audio_transformer_amd.urbansound_classifier import native_cross_entropy
    ref, nat = _train_pair(n_mels, T, seed=7 + T)
    x = torch.from_numpy(wts.unit_variates("cls.xt", batch * n_mels * T, 5).reshape(batch, n_mels, T).astype(np.float32)) * 2.0 - 4.0
    y = torch.tensor([(3 * i + 1) % 10 for i in range(batch)])
    want_loss = torch.nn.CrossEntropyLoss()(ref(x), y)
    want_loss.backward()
    got_loss = native_cross_entropy(nat(x.cuda()), y.cuda())
    got_loss.backward()
    assert abs(float(got_loss.detach()) - float(want_loss.detach())) < 1e-4
    want = dict(ref.named_parameters())
    checked = 0
    for name, p in nat.named_parameters():
        assert p.grad is not None, name
        g, w = p.grad.cpu(), want[name].grad
        scale = float(w.abs().max())
        assert scale > 0, name
        err = float((g - w).abs().max())
        assert err <= 2e-4 * scale + 1e-7, (name, err, scale)
        checked += 1
    assert checked == len(want)


def test_reduction_operators_match_torch():
    from mlx8_ws_audio_transformer_amd import urbansound_classifier as uc
    g = torch.Generator().manual_seed(3)
    for M, d in [(1, 4), (700, 128), (1030, 10), (513, 1280)]:
        a = torch.randn(M, d, generator=g).cuda()
        np.testing.assert_allclose(uc._column_sums(a).cpu().numpy(), a.double().sum(0).float().cpu().numpy(), rtol=2e-5, atol=2e-5)
    x = (torch.randn(700, 128, generator=g) * 3 + 1).cuda().requires_grad_(True)
    gam, bet = torch.randn(128, generator=g).cuda().requires_grad_(True), torch.randn(128, generator=g).cuda().requires_grad_(True)
    dy = torch.randn(700, 128, generator=g).cuda()
    uc._LayerNorm.apply(x, gam, bet, 1e-5).backward(dy)
    got = (x.grad.clone(), gam.grad.clone(), bet.grad.clone())
    x64, g64, b64 = (t.detach().double().requires_grad_(True) for t in (x, gam, bet))
    torch.nn.functional.layer_norm(x64, (128,), g64, b64, 1e-5).backward(dy.double())
    for a, b in zip(got, (x64.grad, g64.grad, b64.grad)):
        np.testing.assert_allclose(a.cpu().numpy(), b.float().cpu().numpy(), rtol=1e-4, atol=2e-4)


def test_train_transformer_learns_a_separable_toy_problem_and_matches_one_adam_step():
    """`train_transformer` over the native operators: (1) after ONE batch the updated weights equal a torch Adam step on the reference
    module (dropout 0); (2) with the reference's dropout 0.1 the loss on a two-class toy set drops well below ln 2."""
    from mlx8_ws_audio_transformer_amd.urbansound_classifier import TransformerUrbanSound8KClassifier, train_transformer
    ref, nat = _train_pair(64, 126, seed=11)
    x = torch.from_numpy(wts.unit_variates("cls.xa", 4 * 64 * 126, 9).reshape(4, 64, 126).astype(np.float32)) * 2.0 - 4.0
    y = torch.tensor([0, 3, 3, 9])
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    torch.nn.CrossEntropyLoss()(ref(x), y).backward(); opt.step()
    train_transformer([(x, y)], model=nat, epochs=1, lr=1e-3)
    want = dict(ref.named_parameters())
    for name, p in nat.named_parameters():
        # Adam's first step moves every weight by ~lr * sign(grad): compare the moved weights where the gradient is above noise level
        # (the key bias has an analytically zero gradient -- softmax is invariant to it -- so its update is the sign of rounding noise)
        g = want[name].grad.abs()
        live = g > 1e-3 * g.max()
        diff = (p.detach().cpu() - want[name].detach()).abs()[live]
        assert diff.numel() > 0 and float(diff.max()) < 2.5e-4, (name, float(diff.max()))

    torch.manual_seed(0)
    model = TransformerUrbanSound8KClassifier(n_classes=2, n_mels=64).cuda()
    gen = torch.Generator().manual_seed(1)
    def batch():
        yb = torch.randint(0, 2, (16,), generator=gen)
        xb = torch.randn(16, 64, 62, generator=gen) * 0.5 - 4.0
        xb[yb == 1, 10:20, :] += 2.0                     # class 1: a brighter band of mel bins
        return xb, yb
    data = [batch() for _ in range(40)]
    model, losses = train_transformer(data, model=model, epochs=3, lr=1e-3)
    assert losses[0] > losses[-1] and losses[-1] < 0.2, losses
    model.eval()
    xb, yb = batch()
    assert float((model(xb.cuda()).argmax(-1).cpu() == yb).float().mean()) >= 0.9


@pytest.mark.parametrize("B,H,S,p", [(2, 3, 70, 0.25), (1, 4, 127, 0.1), (3, 2, 64, 0.5)])
def test_attention_probability_dropout_matches_autograd_with_the_same_mask(B, H, S, p):
    """`nn.TransformerEncoderLayer(dropout=p)` drops attention weights in train() (spectrogram.py:977-985).  The native attention pair
    regenerates a seeded keep mask in its forward and both backward kernels; `attention_keep_mask` restates the mask on the host, and with that very
    mask torch autograd must give the same output and the same dq / dk / dv."""
    from mlx8_ws_audio_transformer_amd.urbansound_classifier import _Attention, attention_keep_mask
    seed = 123456789 + S
    qkv = torch.from_numpy(wts.unit_variates("drop.qkv", B * S * 3 * H * 64, 11).reshape(B * S, 3 * H * 64).astype(np.float32)).cuda().requires_grad_(True)
    dout = torch.from_numpy(wts.unit_variates("drop.do", B * S * H * 64, 12).reshape(B * S, H * 64).astype(np.float32)).cuda()
    o = _Attention.apply(qkv, B, H, S, p, seed)
    o.backward(dout)
    mask = attention_keep_mask(seed, B, H, S, S, p)
    frac = float((mask > 0).float().mean())
    assert abs(frac - (1 - p)) < 0.03, frac                                        # the mask really drops ~p of the weights
    ref_in = qkv.detach().cpu().double().requires_grad_(True)
    q, k, v = (ref_in.reshape(B, S, 3, H, 64)[:, :, i].permute(0, 2, 1, 3) for i in range(3))       # [B, H, S, 64]
    P = torch.softmax(q @ k.transpose(2, 3) * 0.125, dim=-1) * mask.double()
    want = (P @ v).permute(0, 2, 1, 3).reshape(B * S, H * 64)
    want.backward(dout.cpu().double())
    assert float((o.detach().cpu().double() - want.detach()).abs().max()) < 2e-5
    g, w = qkv.grad.cpu().double(), ref_in.grad
    assert float((g - w).abs().max()) < 2e-5 * max(1.0, float(w.abs().max()))
    # p = 0 is the plain kernel pair, bit for bit
    assert torch.equal(_Attention.apply(qkv.detach(), B, H, S, 0.0, seed), _Attention.apply(qkv.detach(), B, H, S))


def test_train_mode_applies_attention_dropout_and_eval_caches_packed_weights():
    """train(): with dropout > 0 two forwards differ (fresh seeds per layer and step, recorded in `last_attention_seeds`) and `torch.manual_seed` reproduces
    a step; eval(): the weights are packed once per parameter version -- a second call reuses the handles, an in-place update re-packs."""
    _, nat = _train_pair(64, 126, seed=3, dropout=0.3)
    x = torch.from_numpy(wts.unit_variates("cls.xd", 2 * 64 * 126, 9).reshape(2, 64, 126).astype(np.float32)).cuda() * 2.0 - 4.0
    torch.manual_seed(5); a = nat(x).detach().clone(); seeds_a = list(nat.last_attention_seeds)
    b = nat(x).detach().clone()
    torch.manual_seed(5); c = nat(x).detach().clone()
    assert len(seeds_a) == 2 and all(s > 0 for s in seeds_a) and seeds_a[0] != seeds_a[1]
    assert not torch.equal(a, b) and torch.equal(a, c)
    nat.eval()
    y0 = nat(x); handles = {k: id(v[1]) for k, v in nat._packed.items.items()}
    y1 = nat(x)
    assert torch.equal(y0, y1) and handles == {k: id(v[1]) for k, v in nat._packed.items.items()} and len(handles) >= 8
    with torch.no_grad():
        nat.head[3].weight.mul_(2.0)
    y2 = nat(x)
    assert not torch.equal(y0, y2) and id(nat._packed.items[id(nat.head[3].weight)][1]) != handles[id(nat.head[3].weight)]
