"""Oracle pinning on CPU: the data-independent identities the reference's own
tests assert, re-run on numpy-generated data, plus
independent cross-checks of the restated third-party formulas."""
import math

import numpy as np
import pytest
import torch

from nnx_ppo_amd import random as keys
from nnx_ppo_amd.envs import DummyCounterEnv, MockEnv
from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper
from oracle import networks as on
from oracle import ppo as op

D = torch.float64


def _mlp(rng, sizes, act="relu", last_act=None):
    layers = []
    for i, (a, b) in enumerate(zip(sizes[:-1], sizes[1:])):
        lim = math.sqrt(3.0 / a)
        w = rng.uniform(-lim, lim, size=(a, b))
        layers.append(on.Dense(w, np.zeros(b), act if i < len(sizes) - 2 else last_act))
    return layers


def _actor_critic(rng, obs, act_dim, hidden=(16, 16), normalize=True, seed=3):
    actor = _mlp(rng, [obs, *hidden, 2 * act_dim])
    critic = on.Sequential(_mlp(rng, [obs, *hidden, 1]))
    sampler = on.NormalTanhSampler(seed, entropy_weight=1e-2, min_std=0.1)
    ad = on.PPOAdapter(on.Sequential([*actor, sampler]), critic)
    return on.Sequential([on.Normalizer(obs), ad]) if normalize else ad


# ---- normalizer_test.py:33-40, 42-65, 97-153, 156-177 ------------------------------
def test_normalizer_initial_is_div_by_ten():
    n = on.Normalizer(4)
    x = torch.tensor(np.random.default_rng(0).normal(size=(8, 4)))
    out = n((), x).output
    assert torch.allclose(out, x / 10.0)


def test_normalizer_welford_matches_moments_and_counter():
    rng = np.random.default_rng(1)
    n = on.Normalizer(3)
    chunks = [rng.normal(2.0, 3.0, size=(7, 5, 3)) for _ in range(4)]
    for c in chunks:
        n.update_statistics(torch.tensor(c))
    data = np.concatenate([c.reshape(-1, 3) for c in chunks])
    assert float(n.counter) == data.shape[0]
    assert np.allclose(n.mean.numpy(), data.mean(0), atol=1e-5)
    std = torch.sqrt(n.M2 / n.counter).numpy()
    assert np.allclose(std, data.std(0), atol=1e-5)
    x = torch.tensor(rng.normal(size=(6, 3)))
    out = n((), x).output.numpy()
    assert np.allclose(out, (x.numpy() - data.mean(0)) / data.std(0), atol=1e-5)


def test_normalizer_call_never_mutates_stats():
    n = on.Normalizer(3)
    n.update_statistics(torch.tensor(np.random.default_rng(2).normal(size=(4, 4, 3))))
    before = (n.mean.clone(), n.M2.clone(), n.counter.clone())
    for _ in range(3):
        n((), torch.ones(5, 3, dtype=D))
    assert torch.equal(before[0], n.mean) and torch.equal(before[1], n.M2)
    assert torch.equal(before[2], n.counter)


def test_normalizer_pytree_leaves_independent():
    rng = np.random.default_rng(3)
    n = on.Normalizer({"a": 2, "b": (3,)})
    xa, xb = rng.normal(1, 2, size=(5, 6, 2)), rng.normal(-3, 0.5, size=(5, 6, 3))
    n.update_statistics({"a": torch.tensor(xa), "b": torch.tensor(xb)})
    assert float(n.counter) == 30
    assert np.allclose(n.mean["a"].numpy(), xa.reshape(-1, 2).mean(0), atol=1e-6)
    assert np.allclose(n.mean["b"].numpy(), xb.reshape(-1, 3).mean(0), atol=1e-6)


# ---- adapter_test.py:61-75: replay reproduces action and loglik -------------------------
def test_replay_reproduces_action_and_loglik():
    rng = np.random.default_rng(4)
    net = _actor_critic(rng, 5, 2, normalize=False)
    st = net.initialize_state(6)
    x = torch.tensor(rng.normal(size=(6, 5)))
    with torch.no_grad():
        o1 = net(st, x)
        o2 = net(st, x, o1.rollout_extras)
    assert torch.allclose(o1.output.actions, o2.output.actions)
    assert torch.allclose(o1.output.loglikelihoods, o2.output.loglikelihoods)


# ---- sampler formulas vs torch.distributions (independent statement) ------------------------
def test_sampler_loglik_and_entropy_formulas():
    rng = np.random.default_rng(5)
    B, A = 7, 3
    ms = torch.tensor(rng.normal(size=(B, 2 * A)))
    s = on.NormalTanhSampler(9, entropy_weight=0.5, min_std=0.1, std_scale=1.3)
    eps = torch.tensor(rng.normal(size=(B, A)))
    eps2 = torch.tensor(rng.normal(size=(B, A)))
    s.noise_override = lambda b, a: (eps, eps2)
    out = s((), ms)
    mean, sp = ms[:, :A], ms[:, A:]
    std = (torch.nn.functional.softplus(sp) + 0.1) * 1.3
    z = mean + std * eps
    assert torch.allclose(out.rollout_extras, z)
    base = torch.distributions.Normal(mean, std)
    # log p(tanh z) = log N(z) - log(1 - tanh(z)^2)
    ll = (base.log_prob(z) - torch.log1p(-torch.tanh(z) ** 2)).sum(-1)
    assert torch.allclose(out.output["log_likelihood"], ll, atol=1e-9)
    z2 = mean + std * eps2
    ent = (base.entropy() + torch.log1p(-torch.tanh(z2) ** 2)).sum(-1)
    assert torch.allclose(out.regularization_loss, -0.5 * ent, atol=1e-9)
    s.deterministic = True
    assert torch.allclose(s((), ms).rollout_extras, mean)


# ---- optimiser formulas vs torch.optim (independent implementation) ---------------------------
@pytest.mark.parametrize("wd", [None, 0.01])
def test_adam_matches_torch_optim(wd):
    rng = np.random.default_rng(6)
    p0 = [rng.normal(size=(4, 3)), rng.normal(size=(3,))]
    mine = [torch.tensor(p, dtype=D) for p in p0]
    ref = [torch.tensor(p, dtype=D, requires_grad=True) for p in p0]
    opt = op.Adam(mine, lr=1e-3, weight_decay=wd)
    topt = (torch.optim.Adam(ref, lr=1e-3, eps=1e-8) if wd is None
            else torch.optim.AdamW(ref, lr=1e-3, eps=1e-8, weight_decay=wd))
    for _ in range(5):
        gs = [torch.tensor(rng.normal(size=p.shape), dtype=D) for p in p0]
        opt.update(gs)
        for r, g in zip(ref, gs):
            r.grad = g.clone()
        topt.step()
    for a, b in zip(mine, ref):
        assert torch.allclose(a, b.detach(), atol=1e-12)


def test_global_norm_clip():
    p = [torch.zeros(3, dtype=D)]
    opt = op.Adam(p, lr=1.0, gradient_clipping=1.0)
    g = torch.tensor([3.0, 4.0, 0.0], dtype=D)
    opt.update([g])  # clipped to norm 1 -> first adam step is -lr*sign-ish
    assert torch.allclose(p[0], torch.tensor([-1.0, -1.0, 0.0], dtype=D), atol=1e-6)


# ---- rollout_test.py:121-192: carry reset in lock-step with env reset ---------------------------
class _CounterNet(on.Module):
    def __call__(self, state, obs, extras=None):
        c = state["counter"] + 1
        f = c.to(D)
        return on.Out({"counter": c}, on.PPOOut(f[:, None], torch.ones_like(f), torch.ones_like(f)),
                      torch.zeros((), dtype=D), {}, None)

    def initialize_state(self, b):
        return {"counter": torch.zeros(b, dtype=torch.int64)}

    def reset_state(self, prev):
        return {"counter": torch.zeros_like(prev["counter"])}


def test_dummy_counter_lock_step():
    N, T = 256, 30
    env = DummyCounterEnv()
    net = _CounterNet()
    es = env.reset(keys.split(keys.key(0), N))
    ns, es2, ro = op.unroll_env(env, es, net, net.initialize_state(N), T,
                                keys.split(keys.key(1), (T, N)))
    assert float(ro.rewards.sum()) == T * N
    nd = int(ro.done.sum())
    assert 2 * N <= nd < 10 * N
    assert ro.obs.shape == (T, N, 1) and ro.done.dtype == torch.bool


# ---- ppo_test.py:38-62 + 340-349: steps, finiteness, normaliser counter ---------------------------
def test_ppo_step_counts_and_finite():
    rng = np.random.default_rng(7)
    N, T = 16, 6
    env = EpisodeWrapper(MockEnv(5, 1, max_steps=4), 1000)
    net = _actor_critic(rng, 5, 1)
    ts = op.new_training_state(env, net, N, 18, keys)
    for k in range(1, 3):
        ts, info = op.ppo_step(env, ts, N, T, 0.95, 0.99, 0.2, True, 2, 2, keys)
        assert ts.steps_taken == k * N * T
        assert float(net.layers[0].counter) == k * N * T
        for name in ("actor", "critic", "regularization"):
            assert info[name].shape == (4,) and torch.isfinite(info[name]).all()
    ro = info["rollout"]
    assert int(ro.done.sum()) > 0  # resets happened
    sc = ts.env_states.info["step_counter"]
    assert int(sc.max()) <= 1000 and int(sc.max()) > 0  # episode_wrapper_test.py:31-57
