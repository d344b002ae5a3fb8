"""`envs.VmapEnv` lifts single-env playground-style envs (the reference's convention,
`rollout.py:21,39`, `docs/reference/batching.rst:70-94`) to the batched convention of this
build: the lifted env must behave exactly like the hand-batched twin, leaf for leaf.
CPU here; the GPU twin (rollout + ppo_step through the kernels) is in
tests/test_vmap_env_gpu.py."""
import torch

from nnx_ppo_amd import random as keys
from nnx_ppo_amd.envs import DummyCounterEnv, MoveToCenterEnv, VmapEnv
from nnx_ppo_amd.tree import tree_leaves

from dummies import SingleDummyCounterEnv, SingleMoveToCenterEnv


def _same(a, b):
    la, lb = tree_leaves(a), tree_leaves(b)
    assert len(la) == len(lb)
    for x, y in zip(la, lb):
        assert x.shape == y.shape, (x.shape, y.shape)
        assert torch.equal(x.to(y.dtype), y), (x, y)


def test_vmapped_dummy_counter_equals_batched():
    n = 17
    rng = keys.split(keys.key(3), n)
    lifted, batched = VmapEnv(SingleDummyCounterEnv()), DummyCounterEnv()
    s, r = lifted.reset(rng), batched.reset(rng)
    _same(s, r)
    assert s.done.shape == (n,) and s.obs.shape == (n, 1)
    for t in range(1, 12):
        a = torch.full((n, 1), float(t))
        s, r = lifted.step(s, a), batched.step(r, a)
        _same(s, r)
    assert lifted.observation_size == 1 and lifted.action_size == 1


def test_vmapped_move_to_center_equals_batched():
    n = 9
    rng = keys.split(keys.key(8), n)
    lifted = VmapEnv(SingleMoveToCenterEnv(1.0, 1.5))
    batched = MoveToCenterEnv(1.0, 1.5)
    s, r = lifted.reset(rng), batched.reset(rng)
    _same(s, r)
    g = torch.Generator().manual_seed(0)
    for _ in range(6):
        a = torch.randn(n, 2, generator=g)
        s, r = lifted.step(s, a), batched.step(r, a)
        _same(s, r)


def test_episode_wrapper_over_a_lifted_env():
    """The product's (batched) EpisodeWrapper composes with a lifted single env."""
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    n = 8
    rng = keys.split(keys.key(1), n)
    a = EpisodeWrapper(VmapEnv(SingleMoveToCenterEnv(1.0, 50.0)), 6)
    b = EpisodeWrapper(MoveToCenterEnv(1.0, 50.0), 6)
    s, r = a.reset(rng), b.reset(rng)
    _same(s, r)
    for _ in range(7):
        act = torch.zeros(n, 2)
        s, r = a.step(s, act), b.step(r, act)
        _same(s, r)
    assert bool(s.info["truncated"].any())
