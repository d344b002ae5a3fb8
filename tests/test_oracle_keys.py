"""`oracle/keys.py` — the key scheme restated in numpy uint64, independently of the product —
pinned by the published SplitMix64 outputs, and the product's CPU statement
(`nnx_ppo_amd/random.py`: torch int64 with emulated logical shifts) checked against it bit
for bit.  (`tests/test_keys_gpu.py` checks the HIP kernels against the same restatement.)
Structure pinned: `ppo.py:271,284-294,544-548`, `rollout.py:57-59` (split / fold_in /
permutation call for call); the VALUES of jax.random's threefry streams are parity unpinned."""
import numpy as np
import pytest
import torch

from nnx_ppo_amd import random as rnd
from oracle import keys as ok


def test_splitmix64_known_answers():
    """Reference outputs of SplitMix64 (Steele, Lea & Flood 2014; Vigna's splitmix64.c):
    seed 1234567, seed 1477776061723855037 (the vector the xoshiro crates test against) and
    the first output for seed 0."""
    assert ok.splitmix64_stream(1234567, 5) == [
        6457827717110365317, 3203168211198807973, 9817491932198370423,
        4593380528125082431, 16408922859458223821]
    assert ok.splitmix64_stream(1477776061723855037, 5) == [
        1985237415132408290, 2979275885539914483, 13511426838097143398,
        8488337342461049707, 15141737807933549159]
    assert ok.splitmix64_stream(0, 1) == [0xE220A8397B1DCDAF]


@pytest.mark.parametrize("seed", [0, 1, 17, 2**31 - 1, 2**63 + 12345, -5])
def test_product_cpu_keys_equal_the_restatement(seed):
    k = rnd.key(seed)
    assert torch.equal(k, ok.key(seed))
    assert torch.equal(rnd.split(k), ok.split(k))
    assert torch.equal(rnd.split(k, 7), ok.split(k, 7))
    kids = rnd.split(k, (3, 5))
    assert torch.equal(kids, ok.split(k, (3, 5)))
    assert torch.equal(rnd.split(kids, 4), ok.split(kids, 4))          # batched parents
    a, b = rnd.split2(kids)
    assert torch.equal(torch.stack([a, b], -1), ok.split(kids, 2))
    for data in (0, 1, 3, 2**40 + 9):
        assert torch.equal(rnd.fold_in(k, data), ok.fold_in(k, data))
        assert torch.equal(rnd.fold_in(kids, data), ok.fold_in(kids, data))
    steps = torch.arange(15, dtype=torch.int64).reshape(3, 5) * 977 - 3
    assert torch.equal(rnd.fold_key(kids, steps), ok.fold_key(kids, steps))
    assert torch.equal(rnd.bits(k, (11,)), ok.bits(k, (11,)))
    assert torch.equal(rnd.bits(kids, (2, 3)), ok.bits(kids, (2, 3)))
    assert torch.equal(rnd.randint(kids, (6,), -3, 10), ok.randint(kids, (6,), -3, 10))
    assert torch.equal(rnd.randint(k, (4,), 5, 5), ok.randint(k, (4,), 5, 5))
    assert torch.equal(rnd.uniform(kids, (9,)), ok.uniform(kids, (9,)))
    assert torch.equal(rnd.unit_uniform(kids, (5,)), ok.unit_uniform(kids, (5,)))
    assert torch.equal(rnd.unit_uniform(kids, (5,), fold=steps),
                       ok.unit_uniform(kids, (5,), fold=steps))


@pytest.mark.parametrize("seed,n", [(0, 1), (3, 2), (17, 64), (23, 1000), (99, 4096)])
def test_permutations_equal_the_restatement(seed, n):
    k = rnd.key(seed)
    p = ok.permutation(k, n)
    assert torch.equal(rnd.permutation(k, n), p)
    assert torch.equal(torch.sort(p).values, torch.arange(n))          # a permutation
    assert torch.equal(rnd.permutations(k, 4, n), ok.permutations(k, 4, n))


def test_minibatch_indices_follow_the_reference_structure():
    """ppo.py:284-294: per epoch e, permutation(fold_in(new_key, e), n_envs) cut into
    n_minibatches rows — the product's `minibatch_indices` against the restatement."""
    from nnx_ppo_amd.algorithms.ppo import minibatch_indices

    new_key = ok.split(ok.key(17))[1]
    n_envs, n_epochs, n_mb = 96, 3, 4
    want = torch.cat([ok.permutation(ok.fold_in(new_key, e), n_envs).reshape(n_mb, -1)
                      for e in range(n_epochs)], 0)
    assert torch.equal(minibatch_indices(new_key, n_envs, n_epochs, n_mb), want)


@pytest.mark.parametrize("obs_size", [5, {"position": 8, "velocity": 9}])
def test_product_envs_on_cpu_equal_the_oracle_envs(obs_size):
    """`oracle/envs.py` (MockEnv `mock_env.py:25-63`, EpisodeWrapper
    `episode_wrapper.py:8-44`, restated with the oracle's keys) against the product's env
    classes on CPU tensors: every leaf of the state after reset and after each of 14 steps,
    resets applied the rollout's way (`rollout.py:39-44`) — bit for bit."""
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.tree import tree_leaves
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper
    from oracle import envs as oe
    from oracle import ppo as op

    pe = EpisodeWrapper(MockEnv(obs_size, 1, max_steps=4), 9)
    qe = oe.EpisodeWrapper(oe.MockEnv(obs_size, 1, max_steps=4), 9)
    k0 = ok.split(ok.key(3), 64)
    ps, qs = pe.reset(k0), qe.reset(k0)

    def same(a, b):
        la = [a.obs, a.reward, a.done, a.info["step_counter"], a.info["truncated"],
              a.data["key"], a.data["step_count"]]
        lb = [b.obs, b.reward, b.done, b.info["step_counter"], b.info["truncated"],
              b.data["key"], b.data["step_count"]]
        for x, y in zip(tree_leaves(la), tree_leaves(lb)):
            assert x.dtype == y.dtype and torch.equal(x, y)

    same(ps, qs)
    reset_keys = ok.split(ok.key(4), (14, 64))
    for t in range(14):
        a = torch.zeros(64, 1)
        ps, qs = pe.step(ps, a), qe.step(qs, a)
        same(ps, qs)
        done = qs.done != 0
        ps = op.tree_where(done, pe.reset(reset_keys[t]), ps)
        qs = op.tree_where(done, qe.reset(reset_keys[t]), qs)
        same(ps, qs)


def test_normal_observation_law_option():
    """`MockEnv(obs_law="normal")`: the reference mock's own law (`mock_env.py:43,53`:
    `jax.random.normal`) as an option of the synthetic env.  Box-Muller on the key scheme's
    bits: the product's statement and the oracle's agree to the last ulp of log / cos (not
    bit for bit: that is why "uniform" stays the default), the draws are N(0, 1), and the
    event stream (flags, counters, keys) stays exact."""
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper
    from oracle import envs as oe

    kids = ok.split(ok.key(11), 4096)
    steps = torch.arange(4096, dtype=torch.int64) % 7
    a = rnd.unit_normal(kids, (5,), fold=steps)
    b = ok.unit_normal(kids, (5,), fold=steps)
    assert a.shape == b.shape == (4096, 5) and a.dtype == torch.float32
    assert np.allclose(a.numpy(), b.numpy(), rtol=0, atol=2e-6)
    z = a.numpy().astype(np.float64).ravel()
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1.0) < 0.03
    assert abs((z ** 3).mean()) < 0.08 and abs((z ** 4).mean() - 3.0) < 0.25   # not uniform (1.8)
    with pytest.raises(ValueError):
        MockEnv(5, 1, obs_law="cauchy")

    pe = EpisodeWrapper(MockEnv(5, 1, max_steps=4, obs_law="normal"), 9)
    qe = oe.EpisodeWrapper(oe.MockEnv(5, 1, max_steps=4, obs_law="normal"), 9)
    k0 = ok.split(ok.key(3), 64)
    ps, qs = pe.reset(k0), qe.reset(k0)
    for t in range(6):
        act = torch.zeros(64, 1)
        ps, qs = pe.step(ps, act), qe.step(qs, act)
        assert np.allclose(ps.obs.numpy(), qs.obs.numpy(), rtol=0, atol=2e-6)
        assert torch.equal(ps.done, qs.done)
        assert torch.equal(ps.info["step_counter"], qs.info["step_counter"])
        assert torch.equal(ps.info["truncated"], qs.info["truncated"])
        assert torch.equal(ps.data["step_count"], qs.data["step_count"])
