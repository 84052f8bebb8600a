"""GPU parity of the individual HIP kernels (through the C-ABI) against plain PyTorch fp32 on the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).cuda()


# tolerances: split-bf16 products carry ~2^-16 relative operand error, single bf16 ~2^-8
TOL = {"bf16x3": 2e-5, "bf16": 1.5e-2, "fp16": 2e-3, "fp16x3": 2e-6, "f16f8": 6e-5}


@pytest.mark.parametrize("tile", [0, 64, 128, 256])                    # 0 = chosen from the shape; the others force each block tiling
@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16", "fp16x3", "f16f8"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 192), (1500, 384, 768), (77, 128, 3072),
                                   (2500, 768, 768), (4096, 256, 128), (3000, 2304, 768)])
def test_linear(precision, M, N, K, tile):
    from mlx8_ws_audio_transformer_amd import _lib, ops
    x, w, b = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3)
    _lib.tuning_set("gemm_tile", tile)
    try:
        y = ops.linear(x, w, b, precision)
    finally:
        _lib.tuning_set("gemm_tile", 0)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    err = (y.double() - ref).abs().max().item()
    print(precision, (M, N, K), tile, "max-abs", err, "ref max", ref.abs().max().item())
    assert err < TOL[precision] * max(1.0, ref.abs().max().item()), err


# the persistent 256 x 256 ping-pong kernel (csrc/gemm_pp.h; tuning knob "gemm_pp" = 2 routes awt_op_linear onto it): split-line activations,
# packed weight regions, the continuous K-tile stream across a workgroup's tiles (more tiles than CUs: M = 70000 x N = 512 is 548 tiles on 256 CUs),
# a ragged last row panel, every K-tile count parity the schedule distinguishes (nk = K / 32 = 4, 6, 24, 96)
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 256, 192), (1000, 512, 768), (2500, 768, 768), (3000, 2304, 768), (777, 768, 3072), (70000, 512, 256),
                                   (700, 1280, 1280), (520, 1280, 5120), (300, 5120, 1280)])     # ... and large-v3's widths (nk = 40, 160)
def test_linear_ping_pong(M, N, K):
    from mlx8_ws_audio_transformer_amd import _lib, ops
    x, w, b = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    y_ship = ops.linear(x, w, b, "f16f8")
    _lib.tuning_set("gemm_pp", 2)
    try:
        y = ops.linear(x, w, b, "f16f8")
        y2 = ops.linear(x, w, b, "f16f8")
    finally:
        _lib.tuning_set("gemm_pp", 1)
    err = (y.double() - ref).abs().max().item()
    print((M, N, K), "ping-pong max-abs", err, "shipped kernel", (y_ship.double() - ref).abs().max().item())
    assert err < TOL["f16f8"] * max(1.0, ref.abs().max().item()), err
    assert torch.equal(y, y2)                       # deterministic (no atomics, fixed tile -> workgroup map)


# The 16 x 16 MFMA form of the f16f8 GEMM (gemm_f8s_kernel; tuning knob "gemm_mfma16", default on): the same arithmetic as the 32 x 32 kernel from different
# fragment layouts (16-row weight copies, one scaled MFMA for both cross terms) -- checked against fp64 and against the 32 x 32 kernel on shapes that select the
# 256-wide tile: ragged M, every encoder K (384 / 768 / 1280 / 3072 / 5120), and an asymmetric identity-like case that would expose a transposed or permuted layout
@pytest.mark.parametrize("M,N,K", [(3000, 768, 768), (2999, 2304, 768), (4100, 768, 3072), (6000, 1536, 384), (3000, 1280, 5120), (36000, 3072, 768),
                                   (3000, 1152, 384), (2999, 384, 1536), (5000, 384, 384), (48000, 1152, 384)])    # ... and Whisper-tiny's widths: a partly empty last column tile
def test_linear_f16f8_on_16x16_mfma(M, N, K):
    from mlx8_ws_audio_transformer_amd import _lib, ops
    x, w, b = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    _lib.tuning_set("gemm_tile", 256)
    try:
        y = ops.linear(x, w, b, "f16f8")
        y2 = ops.linear(x, w, b, "f16f8")
        _lib.tuning_set("gemm_mfma16", 0)
        y32 = ops.linear(x, w, b, "f16f8")
    finally:
        _lib.tuning_set("gemm_mfma16", 1)
        _lib.tuning_set("gemm_tile", 0)
    err, err32 = (y.double() - ref).abs().max().item(), (y32.double() - ref).abs().max().item()
    print((M, N, K), "16x16 max-abs", err, "32x32", err32, "between", (y - y32).abs().max().item())
    assert err < TOL["f16f8"] * max(1.0, ref.abs().max().item()), err
    assert torch.equal(y, y2)
    assert (y - y32).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())    # the same products, another fp32 summation order


def test_linear_f16f8_on_16x16_mfma_places_every_element():
    # x = rows of a permutation-like matrix, W with a distinct value per (n, k): a swapped fragment lane, K block or C register shows up as a wrong element
    from mlx8_ws_audio_transformer_amd import _lib, ops
    M, N, K = 4096, 512, 256
    x = torch.zeros(M, K)
    x[torch.arange(M), (torch.arange(M) * 7) % K] = 1.0
    w = ((torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 37) % 1021 - 510) / 1024 + 1e-4     # not fp16-exact: keeps the general kernel
    _lib.tuning_set("gemm_tile", 256)
    try:
        y = ops.linear(x.cuda(), w.cuda(), None, "f16f8").cpu()
    finally:
        _lib.tuning_set("gemm_tile", 0)
    ref = w.t()[(torch.arange(M) * 7) % K]
    assert (y - ref).abs().max().item() < 1e-6


def test_linear_f16f8_seeded_random_shapes_on_every_route():
    # 36 seeded shapes (ragged M down to 1 row, every N % 128 == 0 up to 1536 incl. the widths whose last 256-column tile is partly empty, K % 64 == 0 up to 1536)
    # through the three f16f8 GEMM routes -- automatic, the 128 x 256 16 x 16-MFMA tiles forced, the ping-pong kernel wherever it applies -- against fp64:
    # partial row panels, partial column tiles and the persistent kernel's panel over-read all in one place
    import random
    from mlx8_ws_audio_transformer_amd import _lib, ops
    rnd = random.Random(20260401)
    shapes = [(rnd.choice([1, 7, 127, 128, 129, 255, 257, 1000, 1501, 3000, 4099]), 128 * rnd.randint(1, 12), 64 * rnd.randint(1, 24)) for _ in range(36)]
    worst = 0.0
    for i, (M, N, K) in enumerate(shapes):
        x, w, b = _rand((M, K), 100 + i), _rand((N, K), 200 + i, K ** -0.5), _rand((N,), 300 + i)
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        bound = TOL["f16f8"] * max(1.0, ref.abs().max().item())
        for route in ("auto", "tile256", "ping-pong"):
            try:
                if route == "tile256":
                    _lib.tuning_set("gemm_tile", 256)
                if route == "ping-pong":
                    _lib.tuning_set("gemm_pp", 2)
                y = ops.linear(x, w, b, "f16f8")
            finally:
                _lib.tuning_set("gemm_tile", 0)
                _lib.tuning_set("gemm_pp", 1)
            err = (y.double() - ref).abs().max().item()
            worst = max(worst, err / bound)
            assert torch.isfinite(y).all() and err < bound, (M, N, K, route, err)
    print("worst error / bound over", len(shapes), "shapes x 3 routes:", worst)


def test_tuning_set_rejects_unknown():
    from mlx8_ws_audio_transformer_amd import _lib
    with pytest.raises(_lib.AwtError):
        _lib.tuning_set("gemm_tile", 100)
    with pytest.raises(_lib.AwtError):
        _lib.tuning_set("no_such_key", 1)


def test_linear_identity_with_asymmetric_weight():
    # A = I, asymmetric W: catches a transposed C write (cdna_hip_programming.md §3)
    from mlx8_ws_audio_transformer_amd import ops
    x = torch.eye(128).cuda()
    w = (torch.arange(128 * 128, dtype=torch.float32).reshape(128, 128) % 251 - 125).cuda() / 64
    y = ops.linear(x, w, None, "bf16x3")
    assert torch.equal(y, w.t().contiguous())
    for precision in ("fp16x3", "f16f8"):      # the weights here are multiples of 1/64 below 2: exact in fp16
        assert torch.equal(ops.linear(x, w, None, precision), w.t().contiguous()), precision


@pytest.mark.parametrize("d", [128, 384, 512, 768])
def test_layernorm(d):
    from mlx8_ws_audio_transformer_amd import ops
    x, g, b = _rand((777, d), 4, 3.0) + 1.5, _rand((d,), 5) + 1.0, _rand((d,), 6)
    y = ops.layernorm(x, g, b)
    ref = torch.nn.functional.layer_norm(x, (d,), g, b, 1e-5)
    assert (y - ref).abs().max().item() < 5e-6


# max-abs error bound of softmax(q k^T) v vs float64 on these inputs, per operand precision
# f16f8: q k^T with every cross term (2^-15 per logit), P V as one fp16 product (11-bit P and V: <= 2^-12 |v|_max ~ 1e-3 for N(0, 1) values);
# the kernel variants that keep P V's e4m3 cross terms (attn_shape 1 .. 5) stay within 2e-4
ATT_TOL = {"bf16x3": 1e-4, "fp16x3": 2e-5, "f16f8": 2e-3, "bf16": 4e-2, "fp16": 5e-3}
ATT_TOL_F16F8_CROSS = 2e-4


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16", "fp16x3", "f16f8"])
@pytest.mark.parametrize("B,H,S", [(1, 2, 64), (2, 3, 200), (1, 2, 1500), (1, 1, 129), (3, 2, 257)])
def test_attention(precision, B, H, S):
    from mlx8_ws_audio_transformer_amd import ops
    q, k, v = _rand((B, H, S, 64), 7, 0.35), _rand((B, H, S, 64), 8), _rand((B, H, S, 64), 9)
    o = ops.attention(q, k, v, precision)
    p = torch.softmax(q.double() @ k.double().transpose(2, 3), dim=-1)
    ref = (p @ v.double()).transpose(1, 2).reshape(B, S, H * 64)
    err = (o.double() - ref).abs().max().item()
    print(precision, (B, H, S), "max-abs", err)
    assert err < ATT_TOL[precision], err  # v_exp_f32 ~1 ulp; bf16 rounds q, k, v and P


@pytest.mark.parametrize("shape", [1, 2, 3, 4, 5, 6, 7])
def test_attention_f16f8_workgroup_shapes(shape):
    """Every workgroup shape of the f16f8 attention kernel (4 x 32, 4 x 64, 6 x 32 queries) on a sequence with a tail tile."""
    from mlx8_ws_audio_transformer_amd import _lib, ops
    for S in (333, 64, 100, 128, 200, 1500):          # 6 / 1 / 2 / 2 / 4 / 24 key tiles, with and without a tail tile
        B, H = 2, 2
        q, k, v = _rand((B, H, S, 64), 17, 0.35), _rand((B, H, S, 64), 18), _rand((B, H, S, 64), 19)
        _lib.tuning_set("attn_shape", shape)
        try:
            o = ops.attention(q, k, v, "f16f8")
        finally:
            _lib.tuning_set("attn_shape", 0)
        p = torch.softmax(q.double() @ k.double().transpose(2, 3), dim=-1)
        ref = (p @ v.double()).transpose(1, 2).reshape(B, S, H * 64)
        assert (o.double() - ref).abs().max().item() < (ATT_TOL_F16F8_CROSS if shape <= 5 else ATT_TOL["f16f8"]), S


@pytest.mark.parametrize("precision", ["bf16x3", "fp16x3", "f16f8", "f16f8-pipe"])
def test_attention_online_softmax_rescale_branch(precision):
    # one key per late tile dominates one query's row: forces the running maximum to jump tile after tile
    from mlx8_ws_audio_transformer_amd import _lib, ops
    if precision == "f16f8-pipe":
        _lib.tuning_set("attn_shape", 4)
        precision = "f16f8"
    B, H, S = 1, 1, 448
    q, k, v = _rand((B, H, S, 64), 10, 0.2), _rand((B, H, S, 64), 11), _rand((B, H, S, 64), 12)
    for t, key in enumerate([70, 150, 260, 390]):
        k[0, 0, key] = q[0, 0, 5] * (20.0 + 15 * t)
    o = ops.attention(q, k, v, precision)
    p = torch.softmax(q.double() @ k.double().transpose(2, 3), dim=-1)
    ref = (p @ v.double()).transpose(1, 2).reshape(B, S, 64)
    # logits reach ~170: fp32 ulp of the exp2 argument is ~1.5e-5; f16f8 carries the logits to 2^-16 relative: 170 * 2^-16 = 2.6e-3 in the exponent
    _lib.tuning_set("attn_shape", 0)
    assert (o.double() - ref).abs().max().item() < (3e-4 if precision != "f16f8" else 1.5e-2)


def test_linear_output_larger_than_2g_elements():
    """M x N > 2^31 output elements (the fused decoder cross-attention projection is 96000 x 18432): 64-bit addressing."""
    from mlx8_ws_audio_transformer_amd import ops
    M, N, K = 96000, 24576, 128
    x, w, b = _rand((M, K), 11), _rand((N, K), 12, K ** -0.5), _rand((N,), 13)
    y = ops.linear(x, w, b, "bf16x3")
    rows = torch.tensor([0, 1, 47999, 87380, 87382, M - 1], device="cuda")       # 87381 * 24576 ~ 2^31
    ref = torch.nn.functional.linear(x[rows].double(), w.double(), b.double())
    assert (y[rows].double() - ref).abs().max().item() < TOL["bf16x3"] * max(1.0, ref.abs().max().item())


def test_fp6_experiment_is_not_in_the_shipped_library():
    """The round-2 FP6 (e3m2) cross-term experiment is compiled only with -DAWT_EXPERIMENTAL_F6: the shipped library refuses the mode
    loudly instead of carrying an unreachable kernel."""
    from mlx8_ws_audio_transformer_amd import _lib, ops
    x, w = _rand((128, 64), 31), _rand((256, 64), 32, 0.125)
    with pytest.raises(_lib.AwtError, match="AWT_EXPERIMENTAL_F6"):
        ops.linear(x, w, None, "f16f6")


@pytest.mark.parametrize("M,N,K", [(1000, 256, 64), (1500, 768, 768), (3000, 768, 3072), (700, 384, 1536), (96, 2304, 768)])
def test_linear_f16f8_with_fp16_exact_weights(M, N, K):
    """Weights that are exactly representable in fp16 (what checkpoints stored in half precision hold) have a zero lo plane: the library
    notices at upload time and runs the GEMM with ONE e4m3 cross term (x_lo w_hi) instead of two.  Same accuracy class as the general
    path, which the same call takes for weights that are not fp16-exact."""
    from mlx8_ws_audio_transformer_amd import ops
    x, w, b = _rand((M, K), 41), _rand((N, K), 42, K ** -0.5), _rand((N,), 43)
    w16 = w.half().float()
    assert not torch.equal(w16, w)
    ref16 = x.double() @ w16.double().t() + b.double()
    ref = x.double() @ w.double().t() + b.double()
    y16 = ops.linear(x, w16, b, "f16f8")            # exact weights: one cross term
    y = ops.linear(x, w, b, "f16f8")                # general path
    e16, e = (y16.double() - ref16).abs().max().item(), (y.double() - ref).abs().max().item()
    print((M, N, K), "exact-weight path", e16, "general path", e)
    tol = 1e-4 * max(1.0, (K / 768) ** 0.5)          # outputs are O(1); K = 64 has the largest weights (K^-1/2)
    assert e16 < tol and e < tol and e16 < 1.5 * e + 1e-5


def test_exact_weight_gemm_is_bit_identical_across_block_tiles():
    """The one-cross-term (fp16-exact weights) GEMM on every block-tile configuration: 128 x 128, 128 x 256, 256 x 256 -- same bits."""
    from mlx8_ws_audio_transformer_amd import _lib, ops
    x, w, b = _rand((2000, 768), 51), _rand((768, 768), 52, 768 ** -0.5).half().float(), _rand((768,), 53)
    outs = {}
    try:
        for tile in (256, 128, 512):
            _lib.tuning_set("gemm_tile", tile)
            outs[tile] = ops.linear(x, w, b, "f16f8")
    finally:
        _lib.tuning_set("gemm_tile", 0)
    assert torch.equal(outs[256], outs[128]) and torch.equal(outs[256], outs[512])
    ref = x.double() @ w.double().t() + b.double()
    assert (outs[256].double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("M,N,K", [(1500, 768, 768), (700, 384, 1536), (96, 2304, 768), (3000, 128, 3072)])
def test_linear_fp16x3_with_fp16_exact_weights(M, N, K):
    """Split-fp16 mode on weights that are exactly fp16: the a_hi w_lo product is zero and is not issued (two products per fragment pair).
    The result stays at the mode's accuracy, for every block tile."""
    from mlx8_ws_audio_transformer_amd import _lib, ops
    x, w, b = _rand((M, K), 61), _rand((N, K), 62, K ** -0.5).half().float(), _rand((N,), 63)
    ref = x.double() @ w.double().t() + b.double()
    try:
        for tile in (0, 64, 128, 256):
            _lib.tuning_set("gemm_tile", tile)
            y = ops.linear(x, w, b, "fp16x3")
            err = (y.double() - ref).abs().max().item()
            assert err < 1e-5 * max(1.0, (K / 768) ** 0.5), (tile, err)     # fp32 accumulation noise of O(1) outputs (bias included)
    finally:
        _lib.tuning_set("gemm_tile", 0)


def test_linear_f16f8_activation_planes_larger_than_4_gib():
    """the f16f8 GEMM addresses its activation planes with 32-bit per-lane offsets.  They are relative to the tile's first
    source row (64-bit base in SGPRs), so a plane beyond 4 GiB -- chunk_clips >= 467 of Whisper-small's MLP hidden, 280 of large's -- is
    addressed correctly: rows near the end (byte offsets > 2^32 from the plane base) against fp64."""
    from mlx8_ws_audio_transformer_amd import ops
    M, N, K = 704000, 256, 3072                      # fp16 plane: 704000 x 3072 x 2 B = 4.33 GB
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((M, K), generator=g, device="cuda")
    w = torch.randn((N, K), generator=g, device="cuda") * K ** -0.5
    for exact in (False, True):
        ww = w.half().float() if exact else w
        y = ops.linear(x, ww, None, "f16f8")
        rows = torch.tensor([0, 1, 127, 349524, 349525, 699050, 699051, 699052, 700000, M - 129, M - 2, M - 1], device="cuda")   # 699051 * 6144 B ~ 2^32
        ref = x[rows].double() @ ww.double().t()
        assert (y[rows].double() - ref).abs().max().item() < TOL["f16f8"] * 2 * max(1.0, ref.abs().max().item())
        del y
