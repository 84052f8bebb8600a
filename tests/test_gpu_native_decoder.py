"""GPU: the decoder-side operators (include/awt.h "Decoder-side operators", scope row f1) against torch autograd, and the native
decoder + loss against `WhisperForConditionalGeneration` vectors (tests/golden/decoder.npz) and against the torch decoder's gradients."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mlx8_ws_audio_transformer_amd import weights as wts
from oracle import logmel as oracle_mel
from tests.util import golden, piano_clips_f32

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("M,N,K", [(12, 256, 128), (768, 1000, 768), (40, 768, 3072)])
def test_packed_linear_forward_and_backward_input(M, N, K):
    from mlx8_ws_audio_transformer_amd.native_decoder import PackedLinear
    x, w, b, r = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3), _rand((M, N), 4)
    pl = PackedLinear(w, b)
    Np = pl.Np
    assert Np % 128 == 0 and Np >= N
    y = pl.forward(x)
    ref = F.linear(x.double(), w.double(), b.double())
    assert (y[:, :N].double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    assert float(y[:, N:].abs().max()) == 0.0 if Np > N else True
    rp = F.pad(r, (0, Np - N))
    y2 = pl.forward(x, resid=rp)
    assert (y2[:, :N].double() - (ref + r.double())).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    dy = F.pad(_rand((M, N), 5), (0, Np - N))
    dx = pl.backward_input(dy.contiguous())
    dref = dy[:, :N].double() @ w.double()
    assert (dx.double() - dref).abs().max().item() < 3e-5 * max(1.0, dref.abs().max().item())


@pytest.mark.parametrize("B,H,Lq,Sk,causal,off", [(2, 2, 12, 12, True, 0), (3, 4, 7, 1500, False, 0), (2, 2, 1, 9, True, 8), (1, 2, 130, 130, True, 0),
                                                  (2, 3, 5, 70, False, 0)])
def test_small_attention_forward_and_backward(B, H, Lq, Sk, causal, off):
    from mlx8_ws_audio_transformer_amd import native_decoder as nd
    d = H * 64
    q, k, v, do = _rand((B * Lq, d), 1), _rand((B * Sk, d), 2), _rand((B * Sk, d), 3), _rand((B * Lq, d), 4)
    o, lse = nd.attention_small((q, 0), d, (k, 0), d, (v, 0), d, B, H, Lq, Sk, causal, off)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    qh = qd.view(B, Lq, H, 64).transpose(1, 2); kh = kd.view(B, Sk, H, 64).transpose(1, 2); vh = vd.view(B, Sk, H, 64).transpose(1, 2)
    s = (qh * 0.125) @ kh.transpose(2, 3)
    if causal:
        i, j = torch.arange(Lq, device="cuda")[:, None], torch.arange(Sk, device="cuda")[None, :]
        s = s.masked_fill(j > i + off, float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * Lq, d)
    assert (o.double() - ref).abs().max().item() < 2e-5
    ref.backward(do.double())
    dq, dk, dv = torch.full_like(q, 7.0), torch.full_like(k, 7.0), torch.full_like(v, 7.0)      # every element must be overwritten
    nd.attention_small_backward((q, 0), d, (k, 0), d, (v, 0), d, o, do, lse, (dq, 0), (dk, 0), (dv, 0), B, H, Lq, Sk, causal, off)
    for got, want in ((dq, qd.grad), (dk, kd.grad), (dv, vd.grad)):
        assert (got.double() - want).abs().max().item() < 5e-5 * max(1.0, want.abs().max().item())


def test_small_attention_reads_fused_buffers_in_place():
    """q | k | v as column blocks of one [M, 3 d] matrix (the fused projection's output), gradients written the same way."""
    from mlx8_ws_audio_transformer_amd import native_decoder as nd
    B, H, L = 2, 2, 10
    d = H * 64
    qkv, do = _rand((B * L, 3 * d), 1), _rand((B * L, d), 2)
    o, lse = nd.attention_small((qkv, 0), 3 * d, (qkv, d), 3 * d, (qkv, 2 * d), 3 * d, B, H, L, L, True, 0)
    o2, _ = nd.attention_small((qkv[:, :d].contiguous(), 0), d, (qkv[:, d:2 * d].contiguous(), 0), d, (qkv[:, 2 * d:].contiguous(), 0), d, B, H, L, L, True, 0)
    assert torch.equal(o, o2)
    dqkv = torch.empty_like(qkv)
    nd.attention_small_backward((qkv, 0), 3 * d, (qkv, d), 3 * d, (qkv, 2 * d), 3 * d, o, do, lse, (dqkv, 0), (dqkv, d), (dqkv, 2 * d), B, H, L, L, True, 0)
    assert torch.isfinite(dqkv).all() and float(dqkv.abs().max()) > 0


def test_cross_entropy_gelu_layernorm_and_embedding_ops():
    from mlx8_ws_audio_transformer_amd import native_decoder as nd
    M, vocab, ld = 37, 1000, 1024
    logits = F.pad(_rand((M, vocab), 1, 3.0), (0, ld - vocab), value=50.0).contiguous()      # padding columns hold junk: they must be ignored
    labels = torch.randint(0, vocab, (M,), generator=torch.Generator().manual_seed(2)).cuda()
    labels[::5] = -100
    loss, dlogits = nd.cross_entropy(logits, labels, vocab)
    z = logits[:, :vocab].double().requires_grad_(True)
    ref = F.cross_entropy(z, labels, ignore_index=-100)
    ref.backward()
    assert abs(float(loss) - float(ref.detach())) < 1e-5
    assert (dlogits[:, :vocab].double() - z.grad).abs().max().item() < 1e-7 and float(dlogits[:, vocab:].abs().max()) == 0.0
    x, dy = _rand((50, 256), 3, 2.0), _rand((50, 256), 4)
    xd = x.double().requires_grad_(True)
    F.gelu(xd).backward(dy.double())
    assert (nd.gelu(x).double() - F.gelu(x.double())).abs().max().item() < 1e-6
    assert (nd.gelu_backward(x, dy).double() - xd.grad).abs().max().item() < 2e-6
    # the one-transcendental erf GELU (common.h gelu_erf: erfc as 2^(poly)) over its whole domain: a dense grid through the fitted range, the tangent continuation
    # beyond |x| / sqrt 2 = 4.2, and magnitudes where the negative branch has to underflow to zero rather than leave a residue
    grid = torch.cat([torch.linspace(-12, 12, 200001), torch.tensor([0.0, -0.0, 1e-30, -1e-30, 5.94, -5.94, 6.5, -6.5, 40.0, -40.0, 1e4, -1e4, 6e4, -6e4, 3.0])]).cuda()      # 200016 values (the operator takes multiples of 4)
    got, want = nd.gelu(grid).double(), F.gelu(grid.double())
    assert ((got - want).abs() / grid.double().abs().clamp(min=1.0)).max().item() < 2.5e-7
    assert (got[grid < -15] == 0).all() and torch.equal(got[grid > 15].float(), grid[grid > 15])
    g, b, dres = _rand((256,), 5) + 1.0, _rand((256,), 6), _rand((50, 256), 7)
    xd = x.double().requires_grad_(True)
    F.layer_norm(xd, (256,), g.double(), b.double(), 1e-5).backward(dy.double())
    assert (nd.layernorm_backward(dy, x, g).double() - xd.grad).abs().max().item() < 1e-5
    assert (nd.layernorm_backward(dy, x, g, dres=dres).double() - (xd.grad + dres.double())).abs().max().item() < 1e-5


@pytest.mark.parametrize("batch,M,N,K", [(3, 144, 1500, 768), (2, 144, 768, 1500), (5, 10, 64, 64), (2, 1500, 768, 288), (12, 20, 768, 64), (1, 257, 132, 70)])
def test_batched_products_against_fp64(batch, M, N, K):
    """awt_bmm / awt_bmm_pack (csrc/bmm.hip): C_z = A_z B_z^T on the MFMA GEMM, padded shapes, residual in place."""
    from mlx8_ws_audio_transformer_amd import native_decoder as nd
    a, b = _rand((batch, M, K), 1), _rand((batch, N, K), 2, K ** -0.5)
    ref = a.double() @ b.double().transpose(1, 2)
    tol = 3e-5 * max(1.0, ref.abs().max().item())
    pb = nd.PackedBatch(b)
    out = torch.full((batch, M, N), 7.0, device="cuda")
    nd.bmm((a, 0, K, M * K), pb, M, (out, 0, N, M * N))
    assert (out.double() - ref).abs().max().item() < tol
    r0 = _rand((batch, M, N), 3)
    acc = r0.clone()
    nd.bmm((a, 0, K, M * K), pb, M, (acc, 0, N, M * N), resid=(acc, 0, N, M * N))
    assert (acc.double() - (ref + r0.double())).abs().max().item() < tol


def test_batched_products_on_strided_head_views():
    """The per-head use: A_h = columns 64 h .. 64 h + 63 of a row-major [M, H 64] matrix, C_h written into rows (m, h) of [M, H, d] and back."""
    from mlx8_ws_audio_transformer_amd import native_decoder as nd
    M, H, d = 24, 3, 192
    q, w = _rand((M, d), 1), _rand((d, d), 2, d ** -0.5)
    wk = w.view(H, 64, d)
    qt = torch.full((M, H, d), 7.0, device="cuda")
    nd.bmm((q, 0, d, 64), nd.PackedBatch(wk.transpose(1, 2)), M, (qt, 0, H * d, d))
    ref = torch.einsum("mhc,hcj->mhj", q.double().view(M, H, 64), wk.double())
    assert (qt.double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    back = torch.full((M, d), 7.0, device="cuda")
    nd.bmm((qt, 0, H * d, d), nd.PackedBatch(wk), M, (back, 0, d, 64))
    ref2 = torch.einsum("mhj,hcj->mhc", qt.double(), wk.double()).reshape(M, d)
    assert (back.double() - ref2).abs().max().item() < 3e-5 * max(1.0, ref2.abs().max().item())


def test_batched_products_with_k_major_operands():
    """awt_bmm_kmajor / awt_bmm_pack_kmajor: both operands read transposed where they lie, each as two blocks stacked along K
    (the backward's d(enc)_b = [P_b^T | dS_b^T] [d(context)_b ; q~_b])."""
    from mlx8_ws_audio_transformer_amd import native_decoder as nd
    B, R, S, Sp, d = 3, 20, 70, 72, 136
    P, dS = _rand((B * R, Sp), 1), _rand((B * R, Sp), 2)
    dc, qt = _rand((B * R, d), 3), _rand((B * R, d), 4)
    acc0 = _rand((B, S, d), 5)
    ref = acc0.double() + P.view(B, R, Sp)[:, :, :S].double().transpose(1, 2) @ dc.view(B, R, d).double() \
        + dS.view(B, R, Sp)[:, :, :S].double().transpose(1, 2) @ qt.view(B, R, d).double()
    rhs = nd.PackedBatch(kmajor=((dc, 0), (qt, 0), R, d, R * d, B, d, 2 * R))
    acc = acc0.clone()
    nd.bmm(None, rhs, S, (acc, 0, d, S * d), resid=(acc, 0, d, S * d), a_kmajor=((P, 0), (dS, 0), R, Sp, R * Sp))
    assert (acc.double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    # one block per side, strided per-head views (the value adapter's d(merged weight)_h = da2_h^T context_h)
    M, H = 24, 2
    da2, c = _rand((M, H * 64), 6), _rand((M, H, d), 7)
    dw = torch.full((H * 64, d), 7.0, device="cuda")
    nd.bmm(None, nd.PackedBatch(kmajor=((c, 0), None, M, H * d, d, H, d, M)), 64, (dw, 0, d, 64 * d), a_kmajor=((da2, 0), None, M, H * 64, 64))
    ref2 = torch.einsum("mhc,mhj->hcj", da2.double().view(M, H, 64), c.double()).reshape(H * 64, d)
    assert (dw.double() - ref2).abs().max().item() < 3e-5 * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize("rows,cols,ld", [(7, 1500, 1500), (3, 70, 72), (2, 4096, 4096), (5, 1, 4)])
def test_row_softmax_forward_and_backward(rows, cols, ld):
    from mlx8_ws_audio_transformer_amd import native_decoder as nd
    s = _rand((rows, ld), 1, 6.0)
    dp = _rand((rows, ld), 2)
    sd = s[:, :cols].double().requires_grad_(True)
    ref = torch.softmax(sd * 0.125, -1)
    ref.backward(dp[:, :cols].double())
    p = nd.softmax_rows(s.clone(), cols, 0.125)
    assert (p[:, :cols].double() - ref).abs().max().item() < 1e-6
    assert torch.equal(p[:, cols:], s[:, cols:])                         # the padding columns are not touched
    ds = nd.softmax_rows_backward(p, dp.clone(), cols, 0.125)
    assert (ds[:, :cols].double() - sd.grad).abs().max().item() < 1e-6 * max(1.0, sd.grad.abs().max().item())


def test_bmm_rejects_bad_arguments():
    from mlx8_ws_audio_transformer_amd import native_decoder as nd
    a, b = _rand((2, 8, 64), 1), _rand((2, 6, 64), 2)                       # N = 6: not a multiple of 4
    out = torch.zeros((2, 8, 6), device="cuda")
    with pytest.raises(RuntimeError, match="multiples of 4"):
        nd.bmm((a, 0, 64, 8 * 64), nd.PackedBatch(b), 8, (out, 0, 6, 48))
    with pytest.raises(RuntimeError, match="columns"):
        nd.softmax_rows(torch.zeros((2, 5000), device="cuda"), 5000, 1.0)


def _pair(native):
    """WhisperLoRAModel (mini encoder, 2 decoder layers, vocab 512) with the deterministic weights of tests/golden/decoder.npz."""
    from mlx8_ws_audio_transformer_amd.finetune import WhisperLoRAModel
    cfg = wts.config("mini", True)
    model = WhisperLoRAModel(cfg, wts.LoraSpec(r=8, alpha=16.0), decoder_layers=2, vocab=512, max_target_positions=64, native_decoder=native)
    model.config.decoder_start_token_id, model.config.pad_token_id, model.config.eos_token_id = 1, 0, 2
    We = wts.init_encoder_weights(cfg, seed=0, profile="test")
    Wd = wts.init_decoder_weights(cfg.d_model, 2, cfg.ffn, 512, 64, seed=0)
    model.encoder.load_state_dict({k: torch.from_numpy(v) for k, v in We.items()}, strict=False)
    model.decoder.load_state_dict({k: torch.from_numpy(v) for k, v in Wd.items()}, strict=True)
    mel = oracle_mel.whisper_logmel(piano_clips_f32(2), n_samples=2 * cfg.max_source_positions * 160)
    return model, torch.from_numpy(mel).cuda()


def test_native_decoder_loss_logits_and_greedy_tokens_match_reference():
    """fineTune.py's forward (a9) and greedy decoding against WhisperForConditionalGeneration on the same weights, decoder on libawt."""
    G = golden("decoder.npz")
    model, mel = _pair(True)
    model.eval()
    with torch.no_grad():
        out = model(input_features=mel, labels=torch.from_numpy(G["labels"]).cuda())
    np.testing.assert_allclose(out.logits.float().cpu().numpy(), G["logits"], rtol=0, atol=2e-3)
    assert abs(float(out.loss) - float(G["loss"])) < 1e-3
    ids = model.generate(mel, max_length=G["greedy_ids"].shape[1]).cpu().numpy()
    np.testing.assert_array_equal(ids, G["greedy_ids"])


@pytest.mark.parametrize("cross_mode", ["kv", "absorbed"])
def test_native_decoder_gradients_match_the_torch_decoder(cross_mode):
    """Adapter gradients of the full step (native encoder backward fed by d(loss)/d(hidden)): native decoder vs stock-PyTorch decoder, with the
    cross-attention on projected keys / values ("kv") and in the absorbed form the training step uses for short label sequences."""
    G = golden("decoder.npz")
    labels = torch.from_numpy(G["labels"]).cuda()
    res = {}
    for native in (False, True):
        model, mel = _pair(native)
        if native:
            model.decoder.cross_mode = cross_mode
            assert model.decoder.absorbed_cross(labels.shape[1], model.encoder.cfg.max_source_positions) == (cross_mode == "absorbed")
        with torch.no_grad():
            for p in model.lora_parameters():
                if p.shape[1] == 8:
                    p.copy_(torch.from_numpy(0.05 * wts.unit_variates("ndec", p.numel(), 1).reshape(p.shape).astype(np.float32)))
        out = model(input_features=mel, labels=labels)
        out.loss.backward()
        res[native] = (float(out.loss.detach()), torch.cat([p.grad.flatten() for p in model.lora_parameters()]).cpu(), out.logits.detach().float().cpu())
    assert abs(res[True][0] - res[False][0]) < 2e-4 * abs(res[False][0])
    assert float((res[True][2] - res[False][2]).abs().max()) < 2e-3
    ref = res[False][1]
    assert float(ref.abs().max()) > 0
    assert float((res[True][1] - ref).abs().max()) < 2e-3 * float(ref.abs().max())


def test_cross_entropy_ignores_only_minus_100():
    """torch.nn.CrossEntropyLoss (what the reference trains with) ignores -100 and RAISES for any other target outside
    [0, vocab).  The wrapper raises too; the C-ABI call alone (which cannot fail without a synchronisation) returns a NaN loss instead of
    silently shrinking the denominator."""
    import ctypes as C
    from mlx8_ws_audio_transformer_amd import _lib, native_decoder as nd
    vocab, ld, M = 1000, 1024, 6
    logits = _rand((M, ld), 3)
    good = torch.tensor([5, -100, 999, 0, -100, 17])
    loss, dl = nd.cross_entropy(logits, good, vocab)
    ref = F.cross_entropy(logits[:, :vocab].double().cpu(), good, ignore_index=-100)
    assert abs(float(loss) - float(ref)) < 1e-5
    for bad in ([5, -100, 1000, 0, -100, 17], [5, -1, 3, 0, -100, 17], [51865, 1, 2, 3, 4, 5]):
        with pytest.raises(IndexError, match="out of bounds"):
            nd.cross_entropy(logits, torch.tensor(bad), vocab)
        lab = torch.tensor(bad, dtype=torch.int64, device="cuda")
        out, dlog, scratch = torch.zeros((), device="cuda"), torch.empty_like(logits), torch.empty(M + 1, device="cuda")
        _lib.check(_lib.lib().awt_op_cross_entropy(_lib.ctx(logits.device), _lib.ptr(logits), _lib.ptr(lab), M, vocab, ld, _lib.ptr(out), _lib.ptr(dlog),
                                                   _lib.ptr(scratch), _lib.stream_handle()))
        assert torch.isnan(out).item()


class _TorchLoRALinear(torch.nn.Module):
    """y = base(x) + (alpha / r) (x A^T) B^T over a frozen nn.Linear: the build-defined adapter restated with torch autograd."""

    def __init__(self, base, A, B, scale):
        super().__init__()
        self.base, self.scale = base, scale
        self.A, self.B = torch.nn.Parameter(A.clone()), torch.nn.Parameter(B.clone())
        self.weight, self.bias = base.weight, base.bias              # the fused cross-K/V path of the torch decoder is not used here

    def forward(self, x):
        return self.base(x) + self.scale * F.linear(F.linear(x, self.A), self.B)


@pytest.mark.parametrize("cross_mode", ["kv", "absorbed"])
@pytest.mark.parametrize("targets", [("q_proj", "v_proj"), ("v_proj",)])
def test_decoder_adapters_forward_and_gradients_match_torch_autograd(targets, cross_mode):
    """Scope row f1, second half ("+ LoRA on decoder"; the reference fine-tunes every decoder parameter, AB/fineTune.py:131,186-199):
    adapters on the decoder's self-attention and cross-attention q_proj / v_proj.  Loss, logits, d(loss)/d(encoder adapters) and
    d(loss)/d(every decoder adapter) of the native path against torch autograd over the stock-PyTorch decoder carrying the same adapters."""
    from mlx8_ws_audio_transformer_amd.finetune import WhisperLoRAModel
    G = golden("decoder.npz")
    labels = torch.from_numpy(G["labels"]).cuda()
    cfg = wts.config("mini", True)
    spec = wts.LoraSpec(r=8, alpha=16.0, targets=targets)
    We = wts.init_encoder_weights(cfg, seed=0, profile="test")
    Wd = wts.init_decoder_weights(cfg.d_model, 2, cfg.ffn, 512, 64, seed=0)
    mel = torch.from_numpy(oracle_mel.whisper_logmel(piano_clips_f32(2), n_samples=2 * cfg.max_source_positions * 160)).cuda()

    def build(native):
        m = WhisperLoRAModel(cfg, wts.LoraSpec(r=8, alpha=16.0), decoder_layers=2, vocab=512, max_target_positions=64, native_decoder=native,
                             decoder_lora=spec if native else None, native_cross_kv=False)
        m.config.decoder_start_token_id, m.config.pad_token_id, m.config.eos_token_id = 1, 0, 2
        m.encoder.load_state_dict({k: torch.from_numpy(v) for k, v in We.items()}, strict=False)
        m.decoder.load_state_dict({k: torch.from_numpy(v) for k, v in Wd.items()}, strict=False)
        with torch.no_grad():
            for p in m.encoder.parameters():
                if p.requires_grad and p.shape[1] == 8:
                    p.copy_(torch.from_numpy(0.05 * wts.unit_variates("declora_enc", p.numel(), 1).reshape(p.shape).astype(np.float32)))
        return m

    nat = build(True)
    nat.decoder.cross_mode = cross_mode
    names = [n for n, _ in nat.decoder.named_parameters() if "lora_" in n]
    assert len(names) == 2 * 2 * len(targets) * 2 and all(p.requires_grad for n, p in nat.decoder.named_parameters() if "lora_" in n)
    assert not any(p.requires_grad for n, p in nat.decoder.named_parameters() if "lora_" not in n)
    vals = {}
    with torch.no_grad():
        for n, p in nat.decoder.named_parameters():
            if n.endswith("lora_B"):                              # non-zero B so that every adapter matrix gets a gradient
                p.copy_(torch.from_numpy(0.05 * wts.unit_variates(n, p.numel(), 3).reshape(p.shape).astype(np.float32)))
            if "lora_" in n:
                vals[n] = p.detach().clone()
    ref = build(False)
    wrapped = {}
    for i, lay in enumerate(ref.decoder.layers):                   # the same adapters on the stock-PyTorch decoder
        for att in ("self_attn", "encoder_attn"):
            for proj in targets:
                key = f"layers.{i}.{att}.{proj}"
                mod = _TorchLoRALinear(getattr(getattr(lay, att), proj), vals[key + ".lora_A"], vals[key + ".lora_B"], spec.scale)
                setattr(getattr(lay, att), proj, mod)
                wrapped[key] = mod
    out_n = nat(input_features=mel, labels=labels)
    out_n.loss.backward()
    out_r = ref(input_features=mel, labels=labels)
    out_r.loss.backward()
    assert abs(float(out_n.loss.detach()) - float(out_r.loss.detach())) < 2e-4 * abs(float(out_r.loss.detach()))
    assert float((out_n.logits.detach().float() - out_r.logits.detach().float()).abs().max()) < 2e-3
    # the adapters really act: the loss differs from the adapter-free reference value
    assert abs(float(out_n.loss.detach()) - float(G["loss"])) > 1e-4
    ge_n = torch.cat([p.grad.flatten() for n, p in nat.encoder.named_parameters() if "lora_" in n])
    ge_r = torch.cat([p.grad.flatten() for n, p in ref.encoder.named_parameters() if "lora_" in n])
    assert float((ge_n - ge_r).abs().max()) < 2e-3 * float(ge_r.abs().max())
    for n, p in nat.decoder.named_parameters():
        if "lora_" not in n:
            assert p.grad is None
            continue
        mod = wrapped[n.rsplit(".", 1)[0]]
        g_ref = (mod.A if n.endswith("lora_A") else mod.B).grad
        assert p.grad is not None and float(g_ref.abs().max()) > 0, n
        assert float((p.grad - g_ref).abs().max()) < 2e-3 * float(g_ref.abs().max()), n
    # greedy decoding runs through the adapters as well (incremental decoding with the cache = the full forward's argmax)
    nat.eval()
    ids = nat.generate(mel, max_length=8)
    with torch.no_grad():
        full = nat(input_features=mel, decoder_input_ids=ids[:, :-1]).logits.argmax(-1)
    assert torch.equal(full[:, -1], ids[:, -1])


def test_trainer_steps_encoder_and_decoder_adapters_together(tmp_path):
    """One flat gradient buffer: [native encoder adapters | decoder adapters]; a few steps lower the loss and both halves move."""
    from mlx8_ws_audio_transformer_amd.finetune import Seq2SeqTrainer, Seq2SeqTrainingArguments, WhisperLoRAModel
    cfg = wts.config("mini", True)
    model = WhisperLoRAModel(cfg, wts.LoraSpec(r=8, alpha=16.0), decoder_layers=1, vocab=512, max_target_positions=64, decoder_lora=wts.LoraSpec(r=4, alpha=8.0))
    mel = torch.from_numpy(oracle_mel.whisper_logmel(piano_clips_f32(4), n_samples=2 * cfg.max_source_positions * 160))
    g = torch.Generator().manual_seed(0)
    labels = torch.randint(3, 500, (4, 6), generator=g)
    tr = Seq2SeqTrainer(args=Seq2SeqTrainingArguments(output_dir=str(tmp_path), learning_rate=5e-3, max_steps=100, predict_with_generate=False), model=model)
    assert tr.bucket.numel > tr.n_native > 0
    before = [p.detach().clone() for p in model.lora_parameters()]
    losses = [tr.training_step({"input_features": mel, "labels": labels}) for _ in range(8)]
    assert losses[-1] < losses[0] - 0.05, losses
    moved = [float((p.detach() - b).abs().max()) for p, b in zip(model.lora_parameters(), before)]
    n_enc = len([1 for n, _ in model.encoder.named_parameters() if "lora_" in n])
    assert max(moved[:n_enc]) > 0 and max(moved[n_enc:]) > 0
    tr.save_model(full=True)
    sd = torch.load(tmp_path / "lora_adapters.pt")["lora"]
    assert any(k.startswith("decoder.") for k in sd) and (tmp_path / "model.safetensors").exists()


def test_absorbed_cross_attention_equals_projected_form_at_whisper_small_shape():
    """The training step's cross-attention at the reference's real shape -- d = 768, 12 heads, S = 1500 encoder positions, L = 12 label rows (H L = 144) --
    in its absorbed form against the projected key / value form: loss, logits and d(loss) / d(encoder states)."""
    from mlx8_ws_audio_transformer_amd.native_decoder import NativeWhisperDecoder
    d, H, S, B, L, vocab = 768, 12, 1500, 3, 12, 1024
    Wd = wts.init_decoder_weights(d, 2, 3072, vocab, 64, seed=0)
    enc0 = _rand((B, S, d), 7)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(3, vocab, (B, L), generator=g).cuda()
    labels = torch.randint(3, vocab, (B, L), generator=g).cuda()
    labels[0, -2:] = -100
    res = {}
    for mode in ("kv", "absorbed"):
        dec = NativeWhisperDecoder(d, 2, H, 3072, vocab, 64).cuda()
        dec.load_state_dict({k: torch.from_numpy(v) for k, v in Wd.items()})
        dec.cross_mode = mode
        enc = enc0.clone().requires_grad_(True)
        loss, logits = dec.loss(ids, labels, enc)
        loss.backward()
        res[mode] = (float(loss.detach()), logits.detach().float(), enc.grad.clone())
    assert abs(res["kv"][0] - res["absorbed"][0]) < 2e-5 * abs(res["kv"][0])
    assert float((res["kv"][1] - res["absorbed"][1]).abs().max()) < 2e-4
    gk, ga = res["kv"][2], res["absorbed"][2]
    assert float(gk.abs().max()) > 0
    assert float((gk - ga).abs().max()) < 2e-4 * float(gk.abs().max())
