"""HIP GAE (mi_gae_f32 through the C ABI) against the oracle and the reference's
known-answer fixture.  Tolerance 1e-6 abs = the reference test's own bound
(ppo_test.py:264); against the fp32 oracle the kernel is bit-exact because it
evaluates the reference's expression order without fp contraction."""
import numpy as np
import pytest
import torch

from oracle import gae as og

pytestmark = pytest.mark.gpu


def _run(dev, r, v, lv, d, tr, gamma, lam, targets=False):
    from nnx_ppo_amd import ops

    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(dev)
    out = ops.gae(t(r, torch.float32), t(v, torch.float32), t(lv, torch.float32),
                  t(d, torch.bool), t(tr, torch.bool), gamma, lam,
                  with_targets=targets)
    torch.cuda.synchronize()
    return out


def test_golden_seed23(dev, golden_dir):
    z = np.load(golden_dir / "gae_seed23.npz")
    g, lam = float(z["gamma"]), float(z["lambda_"])
    adv, tgt = _run(dev, z["rewards"], z["values"][:-1], z["values"][-1], z["done"],
                    z["truncation"], g, lam, targets=True)
    adv = adv.cpu().numpy()
    assert np.max(np.abs(adv.astype(np.float64) - z["advantages"])) < 1e-6
    o32 = og.gae(z["rewards"], z["values"][:-1], z["values"][-1], z["done"],
                 z["truncation"], lam, g, dtype=np.float32)
    assert np.array_equal(adv, o32)
    v32 = z["values"][:-1].astype(np.float32)
    assert np.array_equal(tgt.cpu().numpy(), v32 + o32)


@pytest.mark.parametrize("T,N", [(1, 1), (30, 1024), (30, 4096), (7, 65), (33, 1000), (17, 63)])
def test_shapes_vs_oracle(dev, T, N):
    rng = np.random.default_rng(T * 1000 + N)
    r = rng.normal(size=(T, N)).astype(np.float32)
    v = rng.normal(size=(T, N)).astype(np.float32)
    lv = rng.normal(size=(N,)).astype(np.float32)
    d = rng.random((T, N)) < 0.2
    tr = d & (rng.random((T, N)) < 0.5)
    adv = _run(dev, r, v, lv, d, tr, 0.99, 0.95).cpu().numpy()
    o32 = og.gae(r, v, lv, d, tr, 0.95, 0.99, dtype=np.float32)
    assert np.array_equal(adv, o32)
    o64 = og.gae(r, v, lv, d, tr, 0.95, 0.99)
    assert np.max(np.abs(adv - o64)) < 1e-5


def test_empty(dev):
    adv = _run(dev, np.zeros((0, 8)), np.zeros((0, 8)), np.zeros(8),
               np.zeros((0, 8), bool), np.zeros((0, 8), bool), 0.99, 0.95)
    assert adv.shape == (0, 8)


def test_all_done_all_truncated(dev):
    T, N = 12, 256
    rng = np.random.default_rng(0)
    r = rng.normal(size=(T, N)).astype(np.float32)
    v = rng.normal(size=(T, N)).astype(np.float32)
    lv = rng.normal(size=(N,)).astype(np.float32)
    ones = np.ones((T, N), bool)
    adv = _run(dev, r, v, lv, ones, ones, 0.99, 0.95).cpu().numpy()
    assert np.array_equal(adv, np.zeros_like(adv))
    adv = _run(dev, r, v, lv, ones, ~ones, 0.99, 0.95).cpu().numpy()
    assert np.array_equal(adv, r - v)


def test_large_property(dev):
    """BASELINE-size-and-beyond check through size-independent properties:
    linearity of A in (r, V) for fixed flags, and per-env independence."""
    from nnx_ppo_amd import ops

    T, N = 30, 1 << 18
    g = torch.Generator(device=dev).manual_seed(5)
    r1 = torch.randn(T, N, device=dev, generator=g)
    r2 = torch.randn(T, N, device=dev, generator=g)
    v1 = torch.randn(T, N, device=dev, generator=g)
    v2 = torch.randn(T, N, device=dev, generator=g)
    l1 = torch.randn(N, device=dev, generator=g)
    l2 = torch.randn(N, device=dev, generator=g)
    d = torch.rand(T, N, device=dev, generator=g) < 0.1
    tr = d & (torch.rand(T, N, device=dev, generator=g) < 0.5)
    a1 = ops.gae(r1, v1, l1, d, tr, 0.99, 0.95)
    a2 = ops.gae(r2, v2, l2, d, tr, 0.99, 0.95)
    a12 = ops.gae(r1 + r2, v1 + v2, l1 + l2, d, tr, 0.99, 0.95)
    assert torch.max(torch.abs(a12 - (a1 + a2))).item() < 1e-4
    # a column subset gives the same columns (independence across envs)
    idx = torch.randperm(N, device=dev, generator=g)[:4096]
    sub = ops.gae(r1[:, idx].contiguous(), v1[:, idx].contiguous(), l1[idx].contiguous(),
                  d[:, idx].contiguous(), tr[:, idx].contiguous(), 0.99, 0.95)
    assert torch.equal(sub, a1[:, idx])


@pytest.mark.parametrize("T,N", [(1, 1), (30, 1024), (7, 65), (33, 1000), (5, 64 * 70 + 3)])
def test_fused_statistics(dev, T, N):
    """mi_gae_stats_f32: same advantages as mi_gae_f32 (bit-exact) plus fp64
    (sum A, sum A^2, count) from the same launch; called twice to check that the
    ticket counter in the workspace is left ready for the next launch."""
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(T * 7919 + N)
    r = rng.normal(size=(T, N)).astype(np.float32)
    v = rng.normal(size=(T, N)).astype(np.float32)
    lv = rng.normal(size=(N,)).astype(np.float32)
    d = rng.random((T, N)) < 0.2
    tr = d & (rng.random((T, N)) < 0.5)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(dev)
    args = (t(r, torch.float32), t(v, torch.float32), t(lv, torch.float32), t(d, torch.bool),
            t(tr, torch.bool), 0.99, 0.95)
    plain = ops.gae(*args)
    for _ in range(2):
        adv, stats = ops.gae(*args, with_stats=True)
        assert torch.equal(adv, plain)
        a = plain.cpu().numpy().astype(np.float64)
        s = stats.cpu().numpy()
        assert s[2] == T * N
        assert abs(s[0] - a.sum()) <= 1e-12 * max(1.0, np.abs(a).sum())
        assert abs(s[1] - (a * a).sum()) <= 1e-12 * max(1.0, (a * a).sum())
        # the separate statistics kernel agrees to fp64 rounding
        s2 = ops.adv_stats(plain).cpu().numpy()
        assert np.allclose(s, s2, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("T,N", [(30, 1024), (1, 1), (32, 16384), (7, 100), (30, 65), (12, 64)])
@pytest.mark.parametrize("normalize", [True, False])
@pytest.mark.parametrize("with_reg", [True, False])
def test_fused_gae_loss_equals_two_launches(dev, T, N, normalize, with_reg):
    """mi_gae_ppo_loss_f32 (ppo.py:351-394 + 456-503 in one launch) against
    mi_gae_stats_f32 -> mi_ppo_loss_f32: advantages and both gradients bit for bit (the
    statistics that normalise them are built by the same summation tree), the four loss
    scalars to fp64 summation order.  Called three times: the workspace re-arms itself."""
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(T * 1000 + N)
    t = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt).to(dev)
    for rep in range(3):
        r = t(rng.normal(size=(T, N)))
        v = t(rng.normal(size=(T, N)))
        lv = t(rng.normal(size=(N,)))
        done = t(rng.random((T, N)) < 0.1, torch.bool)
        trunc = done & t(rng.random((T, N)) < 0.5, torch.bool)
        ll_old = t(rng.normal(-1.0, 0.3, size=(T, N)))
        ll_new = ll_old + t(rng.normal(0.0, 0.2, size=(T, N)))
        reg = t(rng.normal(size=(T, N))) if with_reg else None
        if normalize:
            adv, stats = ops.gae(r, v, lv, done, trunc, 0.99, 0.95, with_stats=True)
        else:
            adv, stats = ops.gae(r, v, lv, done, trunc, 0.99, 0.95), None
        g_ll0, g_v0, lo0 = ops.ppo_loss(ll_new.reshape(-1), ll_old.reshape(-1), adv.reshape(-1),
                                        v.reshape(-1), None if reg is None else reg.reshape(-1),
                                        stats, 0.2, 0.7)
        assert ops.gae_ppo_loss_supported(T, N)
        g_ll, g_v, lo, adv1 = ops.gae_ppo_loss(r, v, lv, done, trunc, ll_new, ll_old, reg, 0.99,
                                               0.95, normalize, 0.2, 0.7, want_adv=True)
        assert torch.equal(adv1, adv)
        assert torch.equal(g_ll.reshape(-1), g_ll0), float((g_ll.reshape(-1) - g_ll0).abs().max())
        assert torch.equal(g_v.reshape(-1), g_v0)
        assert torch.allclose(lo, lo0, rtol=1e-6, atol=1e-7), (lo, lo0)
        # deferred: the launch leaves its per-workgroup partials, mi_policy_loss_finalize_f32
        # sums them later in the same order — the same bits as the sum at the launch's tail
        pending: list = []
        lo_d = torch.full((4,), float("nan"), device=dev)
        g_ll2, g_v2, lo_d2, _ = ops.gae_ppo_loss(r, v, lv, done, trunc, ll_new, ll_old, reg, 0.99,
                                                 0.95, normalize, 0.2, 0.7, loss_out=lo_d,
                                                 defer=pending)
        assert lo_d2 is lo_d and len(pending) == 1
        ops.policy_loss_finalize(pending)
        assert not pending and torch.equal(lo_d, lo)
        assert torch.equal(g_ll2, g_ll) and torch.equal(g_v2, g_v)
    assert not ops.gae_ppo_loss_supported(33, 64) and not ops.gae_ppo_loss_supported(8, 16385)


def test_fused_gae_loss_in_ppo_step(dev):
    """The iteration with the one-launch GAE + loss equals the two-launch one."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    outs = []
    for fused in (True, False):
        ppo.FUSED_GAE_LOSS = fused
        try:
            env = EpisodeWrapper(MockEnv(5, 1, max_steps=6), 50)
            net = factories.make_mlp_actor_critic(5, 1, [32, 32], [64], Rngs(2))
            ts = ppo.new_training_state(env, net, 128, 3, 1e-3, device=dev)
            for _ in range(2):
                ts, m = ppo.ppo_step(env, ts, 128, 10, 0.95, 0.99, 0.2, True, False, 2, 2)
            outs.append((ts.optimizer.params.clone(), {k: float(v) for k, v in m.items()}))
        finally:
            ppo.FUSED_GAE_LOSS = True
    assert torch.equal(outs[0][0], outs[1][0])  # same gradients => same parameters
    for k, v in outs[0][1].items():
        assert np.isclose(v, outs[1][1][k], rtol=1e-6, atol=1e-8), k
