"""The 256 x 256-tile NT GEMM with direct-to-LDS loads (csrc/gemm256_bf16.hip) that
`mi_dense_fwd_bf16` / `mi_dense_bwd_dx_bf16` dispatch to for matrix-core-bound layers
(`feedforward.py:42-51` at BASELINE config 3's sizes): against an fp64 evaluation on the same
bf16-rounded operands (1e-4 rel: fp32 accumulation order only), and BIT-IDENTICAL to the
128-row kernel it replaces — reached here by evaluating the same rows in blocks of fewer than
2048 rows, which the dispatch leaves on the old kernel."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D = torch.float64
BF = torch.bfloat16


def _r(a):
    return torch.as_tensor(a, dtype=torch.float32).to(BF).to(D)


SHAPES = [(4096, 512, 512), (4100, 256, 256), (2304, 32, 512), (6000, 512, 256),
          (2048, 64, 384), (3000, 256, 136)]


@pytest.mark.parametrize("M,K,N", SHAPES)
@pytest.mark.parametrize("act", ["relu", "none", "tanh"])
def test_gemm256_forward_and_dx(dev, M, K, N, act):
    from nnx_ppo_amd import _lib, ops

    rng = np.random.default_rng(M + 3 * K + 5 * N)
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(K, N)) / math.sqrt(K)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    code = ops.ACT_CODES[act]
    g = lambda a: torch.as_tensor(a).to(dev)
    x_bf = ops.cast_pad_bf16(g(x))
    w_bf = torch.zeros(K, ops.pad8(N), dtype=BF, device=dev)
    wt_bf = torch.zeros(N, ops.pad8(K), dtype=BF, device=dev)
    ops.weights_to_bf16(g(w), w_bf, wt_bf)

    _, y_bf, _ = ops.dense_fwd_bf16(x_bf, wt_bf, g(b), K, N, code, want_f32=False, want_bf=True)
    z64 = _r(x) @ _r(w) + torch.as_tensor(b, dtype=D)
    y64 = {"none": z64, "relu": torch.relu(z64), "tanh": torch.tanh(z64)}[act]
    got = y_bf[:, :N].float().cpu().to(D)
    # bf16 output: 2^-8 relative rounding on top of the accumulation-order bound
    assert np.allclose(got.numpy(), y64.numpy(), rtol=2.0 ** -7, atol=2e-3), \
        float((got - y64).abs().max())
    # bit-identical to the 128-row kernel (blocks of < 2048 rows stay on it)
    blk = 1000
    old = torch.cat([ops.dense_fwd_bf16(x_bf[i:i + blk].contiguous(), wt_bf, g(b), K, N, code,
                                        want_f32=False, want_bf=True)[1]
                     for i in range(0, M, blk)])
    assert torch.equal(y_bf, old)

    # dX times relu'(prev) (the class the 256-row kernel takes) and without a previous layer
    dz = rng.normal(size=(M, N)).astype(np.float32)
    prev = rng.normal(size=(M, K)).astype(np.float32)
    dz_bf, prev_bf = ops.cast_pad_bf16(g(dz)), ops.cast_pad_bf16(g(prev))
    for p_act, p_bf in ((ops.ACT_RELU, prev_bf), (ops.ACT_NONE, None)):
        _, gx_bf = ops.dense_bwd_dx_bf16(dz_bf, w_bf, p_bf, p_act, K, N, want_f32=False,
                                         want_bf=True)
        gx64 = _r(dz) @ _r(w).t()
        if p_act == ops.ACT_RELU:
            gx64 = gx64 * (_r(prev) > 0).to(D)
        got = gx_bf[:, :K].float().cpu().to(D)
        assert np.allclose(got.numpy(), gx64.numpy(), rtol=2.0 ** -7, atol=2e-3)
        old = torch.cat([ops.dense_bwd_dx_bf16(
            dz_bf[i:i + blk].contiguous(), w_bf,
            None if p_bf is None else p_bf[i:i + blk].contiguous(), p_act, K, N,
            want_f32=False, want_bf=True)[1] for i in range(0, M, blk)])
        assert torch.equal(gx_bf, old)


def test_gemm256_asymmetric_identity(dev):
    """A = I against an asymmetric B pins the operand maps and the swizzle: the output must be
    B's rows exactly, not a transpose or a permutation of them."""
    from nnx_ppo_amd import ops

    M, K, N = 2304, 256, 256
    x = torch.zeros(M, K, device=dev)
    x[:K] = torch.eye(K, device=dev)
    x[K:2 * K] = 2 * torch.eye(K, device=dev)
    w = ((torch.arange(K * N, device=dev, dtype=torch.float32).reshape(K, N) % 251) - 125) / 4
    w_bf = torch.zeros(K, N, dtype=BF, device=dev)
    wt_bf = torch.zeros(N, K, dtype=BF, device=dev)
    ops.weights_to_bf16(w, w_bf, wt_bf)
    _, y_bf, _ = ops.dense_fwd_bf16(ops.cast_pad_bf16(x), wt_bf, None, K, N, ops.ACT_NONE,
                                    want_f32=False, want_bf=True)
    assert torch.equal(y_bf[:K], w.to(BF))
    assert torch.equal(y_bf[K:2 * K], (2 * w).to(BF))
    assert float(y_bf[2 * K:].float().abs().sum()) == 0.0


@pytest.mark.parametrize("M,K,N", [(8192, 512, 512), (12288, 256, 256), (8192, 256, 1024),
                                   (9216, 768, 264), (8192, 520, 512)])
def test_dw256_vs_fp64(dev, M, K, N):
    """dW / db on 256 x 256 tiles (tn256_kernel): fp64 on the same bf16-rounded operands,
    bitwise reproducible, and the grouped form with a thin problem beside it."""
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(M + K + N)
    x = rng.normal(size=(M, K)).astype(np.float32)
    dz = rng.normal(size=(M, N)).astype(np.float32)
    g = lambda a: torch.as_tensor(a).to(dev)
    x_bf, dz_bf = ops.cast_pad_bf16(g(x)), ops.cast_pad_bf16(g(dz))
    gw = torch.zeros(K, N, device=dev)
    gb = torch.zeros(N, device=dev)
    ops.dense_bwd_dw_bf16(x_bf, dz_bf, gw, gb, accumulate=True)
    gw64 = _r(x).t() @ _r(dz)
    gb64 = _r(dz).sum(0)
    s = math.sqrt(M)
    assert np.allclose(gw.cpu().numpy(), gw64.numpy(), rtol=1e-4, atol=2e-5 * s), \
        float((gw.cpu().to(D) - gw64).abs().max())
    assert np.allclose(gb.cpu().numpy(), gb64.numpy(), rtol=1e-4, atol=2e-5 * s)
    gw2 = torch.zeros(K, N, device=dev)
    ops.dense_bwd_dw_bf16(x_bf, dz_bf, gw2, None, accumulate=False)
    assert torch.equal(gw2, gw)
    # grouped: this problem (256-tile kernel) beside a thin head (128-row kernel), one call
    dz2 = rng.normal(size=(M, 3)).astype(np.float32)
    dz2_bf = ops.cast_pad_bf16(g(dz2))
    gw_a, gb_a = torch.zeros(K, N, device=dev), torch.zeros(N, device=dev)
    gw_b, gb_b = torch.zeros(N, 3, device=dev), torch.zeros(3, device=dev)
    ops.dense_bwd_dw_grouped_bf16([(x_bf, dz_bf, gw_a, gb_a), (dz_bf, dz2_bf, gw_b, gb_b)],
                                  accumulate=True)
    if -(-K // 256) * -(-N // 256) >= 4:   # on the 256-tile kernel: the same split either way
        assert torch.equal(gw_a, gw) and torch.equal(gb_a, gb)
    else:                                   # 128-row kernel: the split follows the group's tiles
        assert np.allclose(gw_a.cpu().numpy(), gw64.numpy(), rtol=1e-4, atol=2e-5 * s)
        assert np.allclose(gb_a.cpu().numpy(), gb64.numpy(), rtol=1e-4, atol=2e-5 * s)
    assert np.allclose(gw_b.cpu().numpy(), (_r(dz).t() @ _r(dz2)).numpy(), rtol=1e-4,
                       atol=2e-5 * s)


def test_dw256_asymmetric_identity(dev):
    """x = I rows (twice, so the split-M slabs add up) against an asymmetric dz pins the
    transposing gather and the swizzle of both operands."""
    from nnx_ppo_amd import ops

    M, K, N = 8192, 256, 512
    x = torch.zeros(M, K, device=dev)
    x[:K] = torch.eye(K, device=dev)
    x[4096:4096 + K] = torch.eye(K, device=dev)
    dz = ((torch.arange(M * N, device=dev, dtype=torch.float32).reshape(M, N) % 251) - 125) / 2
    gw = torch.zeros(K, N, device=dev)
    ops.dense_bwd_dw_bf16(ops.cast_pad_bf16(x), ops.cast_pad_bf16(dz), gw, None, accumulate=False)
    want = dz[:K].to(BF).float() + dz[4096:4096 + K].to(BF).float()
    assert torch.equal(gw, want)


GROUPS_128 = [
    # BASELINE configs[1]'s two trunks, one gradient step (M = 30 720)
    (30720, [(5, 64), (64, 64), (64, 64), (64, 64), (64, 2), (5, 256), (256, 256), (256, 1)]),
    # configs[3]: GRU actor (projection and recurrent kernels 64 x 192) beside the 2 x 256 critic
    (30720, [(5, 64), (64, 192), (64, 192), (64, 2), (5, 256), (256, 256), (256, 1)]),
    # ragged widths, M = 32 * odd, a last split shorter than the others
    (32 * 677, [(17, 136), (136, 72), (72, 12), (200, 8)]),
    (96, [(24, 40), (40, 3)]),
]


@pytest.mark.parametrize("M,layers", GROUPS_128)
def test_dw128_dma_kernel_equals_tile_kernel(dev, monkeypatch, M, layers):
    """`tn128_kernel` (csrc/gemm256_bf16.hip): the grouped dW of a training-size step on
    128 x 128 tiles staged by LDS-DMA — same tiles, same split plan, same k order as the
    register-staged tile kernel of gemm_bf16.hip (`MIPPO_DW128_DMA=0`), so the weight and bias
    gradients must come out bit for bit the same; and against fp64 on the same bf16 operands."""
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(M + len(layers))
    probs = []
    for K, N in layers:
        x = torch.as_tensor(rng.normal(size=(M, K)).astype(np.float32)).to(dev)
        dz = torch.as_tensor(rng.normal(size=(M, N)).astype(np.float32)).to(dev)
        probs.append((ops.cast_pad_bf16(x), ops.cast_pad_bf16(dz)))
    res = []
    for flag in ("2", "0"):  # 2: the DMA-staged kernel whatever the split length
        monkeypatch.setenv("MIPPO_DW128_DMA", flag)
        grads = [(torch.zeros(K, N, device=dev), torch.zeros(N, device=dev)) for K, N in layers]
        ops.dense_bwd_dw_grouped_bf16([(xb, zb, gw, gb) for (xb, zb), (gw, gb) in
                                       zip(probs, grads)], accumulate=True)
        ops.flush_pending_slabs()
        res.append(grads)
    for (gw1, gb1), (gw0, gb0), (xb, zb), (K, N) in zip(res[0], res[1], probs, layers):
        assert torch.equal(gw1, gw0) and torch.equal(gb1, gb0), (K, N)
        want = xb[:, :K].to(D).T @ zb[:, :N].to(D)
        assert torch.allclose(gw1.to(D), want, rtol=1e-4, atol=1e-3 * math.sqrt(M)), (K, N)
        assert torch.allclose(gb1.to(D), zb[:, :N].to(D).sum(0), rtol=1e-4,
                              atol=1e-3 * math.sqrt(M)), (K, N)
