"""The weights-stationary trunk kernels (`csrc/trunk_ws.hip`) against the per-tile trunk
kernels of `csrc/mlp_bf16.hip`, which the other tests pin against the oracle: same MFMA
tiles, same accumulation order, same epilogue — every output and every kept bf16 image
must be identical bit for bit (`feedforward.py:42-51` chains, `containers.py:18-39`)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _trunk(dev, dims, seed):
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(seed)
    L = len(dims) - 1
    ffs, bs = [], []
    for l in range(L):
        K, N = dims[l], dims[l + 1]
        w = torch.tensor(rng.normal(size=(K, N)) / math.sqrt(K), dtype=torch.float32, device=dev)
        w_bf = torch.zeros(K, ops.pad8(N), dtype=BF, device=dev)
        wt_bf = torch.zeros(N, ops.pad8(K), dtype=BF, device=dev)
        nf, nb = ops.frag_sizes(K, N)
        ff, fb = torch.zeros(nf, dtype=BF, device=dev), torch.zeros(nb, dtype=BF, device=dev)
        ops.weights_to_bf16_multi([w], [w_bf], [wt_bf], [ff], [fb])
        ffs.append(ff)
        # biases live in a 256-byte aligned arena in the product; keep them aligned here too
        bs.append(torch.tensor(rng.normal(0, 0.3, size=N), dtype=torch.float32, device=dev))
    acts = [ops.ACT_RELU] * (L - 1) + [ops.ACT_NONE]
    return ffs, bs, acts


@pytest.mark.parametrize("dims", [[5, 256, 256, 1], [5, 64, 64, 64, 64, 2], [17, 128, 128, 3],
                                  [5, 64, 2], [32, 256, 16], [9, 128, 128, 128, 12],
                                  [1, 64, 64, 1]])
@pytest.mark.parametrize("M", [1, 63, 64, 1000, 16384, 30720, 31744])
def test_ws_forward_bit_identical_to_tile_kernels(dev, dims, M):
    from nnx_ppo_amd import ops

    ffs, bs, acts = _trunk(dev, dims, seed=M + len(dims))
    assert ops.mlp_ws_supported(dims, acts)
    x = torch.tensor(np.random.default_rng(M).normal(size=(M, dims[0])), dtype=torch.float32,
                     device=dev)
    want, saved0 = ops.mlp_fwd_bf16(x, ffs, bs, dims, acts, train=True)
    got, saved1 = ops.mlp_ws_fwd_bf16(x, ffs, bs, dims, acts, train=True)
    assert torch.equal(got, want), float((got - want).abs().max())
    for l, ((xa, ya), (xb, yb)) in enumerate(zip(saved0, saved1)):
        assert torch.equal(xa, xb), ("input image", l)
        if ya is not None or yb is not None:
            assert torch.equal(ya, yb), ("output image", l)
    inf, none = ops.mlp_ws_fwd_bf16(x, ffs, bs, dims, acts, train=False)
    assert none is None and torch.equal(inf, want)


def test_ws_shape_class(dev):
    from nnx_ppo_amd import ops

    R, N = ops.ACT_RELU, ops.ACT_NONE
    assert ops.mlp_ws_supported([5, 256, 256, 1], [R, R, N])
    assert ops.mlp_ws_supported([5, 64, 64, 64, 64, 2], [R, R, R, R, N])
    assert not ops.mlp_ws_supported([5, 256, 256, 256, 1], [R, R, R, N])   # registers
    assert not ops.mlp_ws_supported([5, 64, 128, 1], [R, R, N])            # unequal widths
    assert not ops.mlp_ws_supported([5, 64, 64, 1], [R, ops.ACT_TANH, N])  # relu only
    assert not ops.mlp_ws_supported([40, 64, 1], [R, N])                   # K0 <= 32
    assert not ops.mlp_ws_supported([5, 64, 17], [R, N])                   # N_out <= 16
    assert not ops.mlp_ws_supported([5, 512, 1], [R, N])
