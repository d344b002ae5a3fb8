"""Per-kernel parity (through the C ABI) against the CPU oracle.
Tolerances: byte/index work bit-exact; fp32 kernels vs the fp64 oracle at the
bound written in each test (normaliser 1e-5 abs = the reference
test's own bound; fp32 kernels <= 1e-5 rel on losses/returns)."""
import math

import numpy as np
import pytest
import torch

from oracle import networks as on
from oracle import philox as oph
from oracle import ppo as op

pytestmark = pytest.mark.gpu
D = torch.float64


def _g(a, dev, dt=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(dev)


# ------------------------------------------------------------------ normaliser
@pytest.mark.parametrize("M,F", [(1, 1), (64, 5), (30 * 4096, 5), (1000, 17), (77, 300), (4096, 2049)])
def test_normalize_fwd(dev, M, F):
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(M + F)
    x = rng.normal(size=(M, F)).astype(np.float32)
    mean = rng.normal(size=F).astype(np.float32)
    m2 = (rng.random(F) * 50).astype(np.float32)
    m2[0] = 0.0  # exercises the epsilon floor
    for cnt in (0.0, 37.0):
        out = ops.normalize_fwd(_g(x, dev), _g(mean, dev), _g(m2, dev), _g([cnt], dev), 1e-6)
        if cnt > 0:
            std = np.sqrt(np.maximum(m2.astype(np.float64) / cnt, 1e-6))
        else:
            std = np.full(F, 10.0)
        want = (x.astype(np.float64) - mean) / std
        assert np.allclose(out.cpu().numpy(), want, rtol=2e-6, atol=1e-6)
        # [x ; x_tail] in one launch (`mi_normalize_fwd_tail_f32`): the same bits, row for row
        k = max(1, M // 3)
        both = torch.empty(M + k, F, dtype=torch.float32, device=dev)
        ops.normalize_fwd_tail(_g(x, dev), _g(x[:k], dev), _g(mean, dev), _g(m2, dev),
                               _g([cnt], dev), 1e-6, both)
        assert torch.equal(both[:M], out) and torch.equal(both[M:], out[:k])
    g = rng.normal(size=(M, F)).astype(np.float32)
    gx = ops.normalize_bwd(_g(g, dev), _g(m2, dev), _g([37.0], dev), 1e-6)
    assert np.allclose(gx.cpu().numpy(), g / np.sqrt(np.maximum(m2 / 37.0, 1e-6)), rtol=2e-6)


@pytest.mark.parametrize("T,B,F", [(30, 4096, 5), (7, 3, 17), (1, 1, 1), (5, 11, 300), (30, 512, 8)])
def test_welford_update_matches_oracle(dev, T, B, F):
    """normalizer_test.py:42-65: after updates, mean/std == moments of all data
    (tol 1e-5), counter == number of samples."""
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(T * B + F)
    mean = torch.zeros(F, device=dev)
    m2 = torch.zeros(F, device=dev)
    cnt = torch.zeros(1, device=dev)
    o = on.Normalizer(F)
    chunks = []
    for k in range(3):
        x = (rng.normal(1.5, 2.0, size=(T, B, F)) + np.arange(F)).astype(np.float32)
        chunks.append(x.reshape(-1, F))
        stats = ops.welford_batch_stats(_g(x, dev), F)
        ops.welford_merge(mean, m2, cnt, stats, advance_counter=True)
        o.update_statistics(torch.tensor(x))
    data = np.concatenate(chunks).astype(np.float64)
    assert float(cnt.item()) == data.shape[0] == float(o.counter)
    assert np.allclose(mean.cpu().numpy(), data.mean(0), atol=1e-5)
    assert np.allclose(mean.cpu().numpy(), o.mean.numpy(), atol=1e-5)
    std = np.sqrt(m2.cpu().numpy() / data.shape[0])
    assert np.allclose(std, data.std(0), atol=1e-5, rtol=1e-5)


# --------------------------------------------------------------------- sampler
def test_philox_matches_oracle(dev):
    from nnx_ppo_amd import ops

    for seed, off, add in [(0, 0, 0), (12345678901234567, 41, 3), (2**63 - 5, 2**40, 7)]:
        st = ops.make_rng_state(seed, dev, off)
        e, e2 = ops.philox_normal(st, add, 5000)
        we, we2 = oph.normal_pair(seed, off + add, 5000)
        assert np.max(np.abs(e.cpu().numpy() - we)) < 5e-6
        assert np.max(np.abs(e2.cpu().numpy() - we2)) < 5e-6
    ops.rng_advance(st, 9)
    assert int(st[1].item()) == 2**40 + 9


@pytest.mark.parametrize("B,A", [(1, 1), (4096, 1), (1000, 6), (30 * 1024, 1), (333, 12)])
@pytest.mark.parametrize("deterministic", [False, True])
def test_sampler_fwd_bwd_vs_oracle(dev, B, A, deterministic):
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(B * 31 + A)
    ms = rng.normal(size=(B, 2 * A)).astype(np.float32)
    kw = dict(min_std=0.1, std_scale=1.2, entropy_weight=0.03)
    seed, off = 99, 5
    st = ops.make_rng_state(seed, dev, off)
    r = ops.tanh_gauss_fwd(_g(ms, dev), None, st, 2, deterministic=deterministic,
                           want_stats=True, **kw)
    o = on.NormalTanhSampler(seed, 0.03, 0.1, 1.2)
    o.offset = off + 2
    o.deterministic = deterministic
    ms64 = torch.tensor(ms, dtype=D, requires_grad=True)
    out = o((), ms64)
    tol = dict(rtol=2e-5, atol=2e-5)
    assert np.allclose(r["raw"].cpu().numpy(), out.rollout_extras.detach().numpy(), **tol)
    assert np.allclose(r["action"].cpu().numpy(), out.output["action"].detach().numpy(), **tol)
    assert np.allclose(r["log_likelihood"].cpu().numpy(),
                       out.output["log_likelihood"].detach().numpy(), rtol=2e-5, atol=1e-4)
    assert np.allclose(r["reg"].cpu().numpy(), out.regularization_loss.detach().numpy(),
                       rtol=2e-5, atol=2e-5)
    assert np.allclose(r["sigma"].cpu().numpy(), out.metrics["sigma"].detach().numpy(), **tol)

    # replay: stored raw action, new parameters, fresh entropy noise; grads vs autograd
    ms2 = (ms + 0.05 * rng.normal(size=ms.shape)).astype(np.float32)
    raw = r["raw"]
    r2 = ops.tanh_gauss_fwd(_g(ms2, dev), raw, st, 3, deterministic=deterministic,
                            want_action=False, want_raw=False, **kw)
    g_ll = rng.normal(size=B).astype(np.float32)
    g_reg = 1.0 / B
    g = ops.tanh_gauss_bwd(_g(ms2, dev), raw, st, 3, _g(g_ll, dev), g_reg, **kw)
    o.offset = off + 3
    m64 = torch.tensor(ms2, dtype=D, requires_grad=True)
    out2 = o((), m64, raw.cpu().to(D))
    obj = (out2.output["log_likelihood"] * torch.tensor(g_ll, dtype=D)).sum() \
        + g_reg * out2.regularization_loss.sum()
    (want,) = torch.autograd.grad(obj, m64)
    assert np.allclose(r2["log_likelihood"].cpu().numpy(),
                       out2.output["log_likelihood"].detach().numpy(), rtol=2e-5, atol=1e-4)
    assert np.allclose(g.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-5)
    # adapter_test.py:61-75: replay with unchanged parameters reproduces loglik
    r3 = ops.tanh_gauss_fwd(_g(ms, dev), raw, st, 4, deterministic=deterministic,
                            want_action=True, want_raw=False, **kw)
    assert torch.allclose(r3["log_likelihood"], r["log_likelihood"], rtol=0, atol=0)
    assert torch.equal(r3["action"], r["action"])


def test_sampler_injected_noise_is_used(dev):
    from nnx_ppo_amd import ops

    B, A = 50, 3
    ms = torch.randn(B, 2 * A, device=dev)
    eps = torch.randn(B, A, device=dev)
    eps2 = torch.randn(B, A, device=dev)
    r = ops.tanh_gauss_fwd(ms, None, None, 0, deterministic=False, eps=eps, eps2=eps2,
                           min_std=0.1, std_scale=1.0, entropy_weight=0.01)
    sigma = torch.nn.functional.softplus(ms[:, A:]) + 0.1
    assert torch.allclose(r["raw"], ms[:, :A] + sigma * eps, atol=1e-6)


# ----------------------------------------------------------------------- dense
@pytest.mark.parametrize("M,K,N", [(1, 1, 1), (4096, 5, 64), (30720, 64, 64), (30720, 64, 2),
                                   (30720, 5, 256), (2048, 256, 256), (2048, 256, 1),
                                   (777, 17, 512), (513, 512, 12), (130, 33, 65)])
@pytest.mark.parametrize("act", ["none", "relu", "tanh", "swish"])
def test_dense_fwd_bwd_vs_oracle(dev, M, K, N, act):
    from nnx_ppo_amd import ops

    if act in ("tanh", "swish") and M > 5000:
        pytest.skip("covered by relu/none at this size")
    rng = np.random.default_rng(M + 7 * K + 13 * N)
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(K, N)) / math.sqrt(K)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    gy = rng.normal(size=(M, N)).astype(np.float32)
    code = ops.ACT_CODES[act]
    xg, wg, bg, gyg = _g(x, dev), _g(w, dev), _g(b, dev), _g(gy, dev)
    if code == ops.ACT_SWISH:
        y, aux = ops.dense_fwd(xg, wg, bg, code, want_preact=True)
    else:
        y = ops.dense_fwd(xg, wg, bg, code)
        aux = y
    x64 = torch.tensor(x, dtype=D, requires_grad=True)
    layer = on.Dense(w, b, None if act == "none" else act)
    y64 = layer((), x64).output
    assert np.allclose(y.cpu().numpy(), y64.detach().numpy(), rtol=1e-5, atol=1e-5)
    gx64, gw64, gb64 = torch.autograd.grad((y64 * torch.tensor(gy, dtype=D)).sum(),
                                            [x64, layer.kernel, layer.bias])
    gx = ops.dense_bwd_dx(gyg, aux, wg, code)
    assert np.allclose(gx.cpu().numpy(), gx64.numpy(), rtol=1e-4, atol=1e-4)
    gw = torch.zeros(K, N, device=dev)
    gb = torch.zeros(N, device=dev)
    ops.dense_bwd_dw(xg, gyg, aux, gw, gb, code, accumulate=True)
    scale = math.sqrt(M)
    assert np.allclose(gw.cpu().numpy(), gw64.numpy(), rtol=1e-4, atol=2e-5 * scale)
    assert np.allclose(gb.cpu().numpy(), gb64.numpy(), rtol=1e-4, atol=2e-5 * scale)
    # accumulate semantics + run-to-run bitwise reproducibility
    gw2 = torch.zeros(K, N, device=dev)
    ops.dense_bwd_dw(xg, gyg, aux, gw2, None, code, accumulate=False)
    assert torch.equal(gw2, gw)
    ops.dense_bwd_dw(xg, gyg, aux, gw2, None, code, accumulate=True)
    assert torch.allclose(gw2, 2 * gw)


def test_dense_asymmetric_layout(dev):
    """A = I against an asymmetric W catches a transposed C write."""
    from nnx_ppo_amd import ops

    K = N = 70
    w = torch.arange(K * N, dtype=torch.float32, device=dev).reshape(K, N)
    y = ops.dense_fwd(torch.eye(K, device=dev), w, None, ops.ACT_NONE)
    assert torch.equal(y, w)
    gx = ops.dense_bwd_dx(torch.eye(N, device=dev), None, w, ops.ACT_NONE)
    assert torch.equal(gx, w.t())


# ------------------------------------------------------------------------ loss
@pytest.mark.parametrize("n", [1, 480, 30720, 61440, 300001])
@pytest.mark.parametrize("normalize", [True, False])
def test_ppo_loss_vs_oracle(dev, n, normalize):
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(n)
    ll_old = rng.normal(-1.0, 0.5, size=n).astype(np.float32)
    ll_new = (ll_old + rng.normal(0, 0.3, size=n)).astype(np.float32)
    adv = rng.normal(0.3, 2.0, size=n).astype(np.float32)
    val = rng.normal(size=n).astype(np.float32)
    reg = rng.normal(size=n).astype(np.float32)
    stats = ops.adv_stats(_g(adv, dev)) if normalize else None
    if normalize:
        s = stats.cpu().numpy()
        assert abs(s[0] - adv.astype(np.float64).sum()) < 1e-6 * n and s[2] == n
    g_ll, g_v, out = ops.ppo_loss(_g(ll_new, dev), _g(ll_old, dev), _g(adv, dev), _g(val, dev),
                                  _g(reg, dev), stats, 0.2, 0.7)
    lln = torch.tensor(ll_new, dtype=D, requires_grad=True)
    v = torch.tensor(val, dtype=D, requires_grad=True)
    a = torch.tensor(adv, dtype=D)
    target = (v + a).detach()
    an = (a - a.mean()) / (a.std(unbiased=False) + 1e-8) if (normalize and n > 0) else a
    ratio = torch.exp(lln - torch.tensor(ll_old, dtype=D))
    actor = -torch.minimum(ratio * an, torch.clamp(ratio, 0.8, 1.2) * an).mean()
    critic = 0.5 * ((v - target) ** 2).mean()
    total = actor + 0.7 * critic
    gl, gv = torch.autograd.grad(total, [lln, v])
    got = out.cpu().numpy()
    if n > 1 or not normalize:
        assert np.allclose(got[0], actor.item(), rtol=2e-5, atol=1e-6)
        assert np.allclose(g_ll.cpu().numpy(), gl.numpy(), rtol=1e-4, atol=1e-7 / max(n, 1) + 1e-9)
    assert np.allclose(got[1], critic.item(), rtol=2e-5, atol=1e-7)
    assert np.allclose(got[2], reg.astype(np.float64).mean(), rtol=2e-5, atol=1e-6)
    clipfrac = (torch.abs(ratio.detach() - 1) > 0.2).double().mean().item()
    assert abs(got[3] - clipfrac) < 2e-4
    # critic gradient: d(0.5 mean((V - sg(V+A))^2))/dV = (V - target)/n, fp32 rounding of V+A
    assert np.allclose(g_v.cpu().numpy(), gv.numpy(), rtol=1e-3, atol=2e-6 / n)


# ------------------------------------------------------------------- optimiser
@pytest.mark.parametrize("clip,wd", [(None, 0.0), (0.5, 0.0), (1e9, 1e-4), (0.5, 0.01)])
def test_adam_vs_oracle(dev, clip, wd):
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(3)
    n = 80579
    p0 = rng.normal(size=n).astype(np.float32)
    p = _g(p0, dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    g = torch.zeros(n, device=dev)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    po = [torch.tensor(p0, dtype=D)]
    o = op.Adam(po, lr=3e-3, gradient_clipping=clip, weight_decay=wd if wd else None)
    for k in range(4):
        gk = (rng.normal(size=n) * 0.01).astype(np.float32)
        ops.begin_grad_step(g, step)
        assert float(g.abs().sum()) == 0.0 and int(step.item()) == k + 1
        g.copy_(_g(gk, dev))
        gn = ops.global_norm(g) if clip is not None else None
        if gn is not None:
            assert abs(gn.item() - np.linalg.norm(gk.astype(np.float64))) < 1e-5
        ops.adam_step(p, g, m, v, step, lr=3e-3, weight_decay=wd, grad_norm=gn,
                      max_norm=clip or 0.0)
        o.update([torch.tensor(gk, dtype=D)])
    assert np.allclose(p.cpu().numpy(), po[0].numpy(), rtol=1e-5, atol=2e-6)


def test_adam_that_opens_the_next_step_equals_begin_plus_adam(dev):
    """`begin_next`: one launch = adam of step t + begin of step t+1; parameters and
    moments bit-identical to the two-launch sequence, gradients left zeroed, step
    counter = completed steps."""
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(11)
    n = 80579
    mk = lambda: [_g(rng.normal(size=n).astype(np.float32), dev)] + \
        [torch.zeros(n, device=dev) for _ in range(3)]
    rng = np.random.default_rng(11)
    pa, ma, va, ga = mk()
    rng = np.random.default_rng(11)
    pb, mb, vb, gb = mk()
    sa = torch.zeros(1, dtype=torch.int64, device=dev)
    sb = torch.zeros(1, dtype=torch.int64, device=dev)
    for k in range(5):
        gk = _g((rng.normal(size=n) * 0.01).astype(np.float32), dev)
        ops.begin_grad_step(ga, sa)
        ga.copy_(gk)
        ops.adam_step(pa, ga, ma, va, sa, lr=3e-3, weight_decay=1e-4)
        assert float(gb.abs().sum()) == 0.0 and int(sb.item()) == k
        gb.copy_(gk)
        ops.adam_step(pb, gb, mb, vb, sb, lr=3e-3, weight_decay=1e-4, begin_next=True)
        assert int(sb.item()) == k + 1 == int(sa.item())
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert float(gb.abs().sum()) == 0.0


# -------------------------------------------------------------------- movement
@pytest.mark.parametrize("dtype,feat", [(torch.float32, (5,)), (torch.float32, ()), (torch.bool, ()),
                                        (torch.int64, (3,)), (torch.float32, (2, 3)),
                                        (torch.uint8, (3,))])
def test_gather_cols_bit_exact(dev, dtype, feat):
    from nnx_ppo_amd import ops

    T, N, L = 7, 200, 64
    g = torch.Generator(device="cpu").manual_seed(0)
    if dtype == torch.bool:
        src = torch.rand(T, N, *feat, generator=g) < 0.5
    elif dtype.is_floating_point:
        src = torch.randn(T, N, *feat, generator=g)
    else:
        src = torch.randint(0, 100, (T, N, *feat), generator=g).to(dtype)
    idx = torch.randperm(N, generator=g)[:L]
    got = ops.gather_cols(src.to(dev), idx.to(dev))
    assert torch.equal(got.cpu(), src[:, idx])


def test_gather_empty_and_full_perm(dev):
    from nnx_ppo_amd import ops

    src = torch.randn(3, 10, 4, device=dev)
    assert ops.gather_cols(src, torch.zeros(0, dtype=torch.int64, device=dev)).shape == (3, 0, 4)
    perm = torch.randperm(10, device=dev)
    inv = torch.argsort(perm)
    assert torch.equal(ops.gather_cols(ops.gather_cols(src, perm), inv), src)


@pytest.mark.parametrize("dtype,feat", [(torch.float32, (64,)), (torch.int64, ()),
                                        (torch.bool, ()), (torch.float32, (3, 2))])
def test_select_rows_bit_exact(dev, dtype, feat):
    from nnx_ppo_amd import ops

    B = 333
    g = torch.Generator(device="cpu").manual_seed(1)
    mk = lambda: (torch.randn(B, *feat, generator=g) * 10).to(dtype)
    a, b = mk(), mk()
    mask = torch.rand(B, generator=g) < 0.3
    got = ops.select_rows(mask.to(dev), a.to(dev), b.to(dev))
    m = mask.reshape(B, *([1] * len(feat)))
    assert torch.equal(got.cpu(), torch.where(m, a, b))
    if feat:
        row = mk()[0]
        got = ops.select_rows(mask.to(dev), row.to(dev), b.to(dev))
        assert torch.equal(got.cpu(), torch.where(m, row.expand_as(b), b))
