"""Host-side logic on CPU: pytree conventions, integer key derivation,
EpisodeWrapper counters/flags, config defaults, callback cadence rule,
synthetic envs' determinism."""
import dataclasses

import numpy as np
import pytest
import torch

from nnx_ppo_amd import random as keys
from nnx_ppo_amd import tree
from nnx_ppo_amd.algorithms import config as cfg
from nnx_ppo_amd.algorithms.ppo import _should_run, minibatch_indices
from nnx_ppo_amd.algorithms.types import LoggingLevel, State
from nnx_ppo_amd.envs import DummyCounterEnv, MockEnv, cheetah_shaped
from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper


def test_tree_conventions():
    t = {"b": torch.ones(2), "a": [torch.zeros(1), None], "c": (torch.ones(3),)}
    leaves = tree.tree_leaves(t)
    assert [x.numel() for x in leaves] == [1, 2, 3]  # sorted keys, None skipped
    m = tree.tree_map(lambda x: x + 1, t)
    assert m["a"][1] is None and torch.equal(m["b"], torch.full((2,), 2.0))
    s = State(data={}, obs=torch.zeros(2), reward=torch.ones(2), done=torch.zeros(2))
    s2 = tree.tree_map(lambda x: x * 2, s)
    assert isinstance(s2, State) and torch.equal(s2.reward, torch.full((2,), 2.0))
    assert s.replace(done=torch.ones(2)).done.sum() == 2
    with pytest.raises(ValueError):
        tree.tree_map(lambda a, b: a, {"x": 1}, {"y": 2})


def test_keys_are_deterministic_and_distinct():
    k = keys.key(17)
    a, b = keys.split(k)[0], keys.split(k)[1]
    assert int(a) != int(b) != int(k)
    assert torch.equal(keys.split(k, (3, 4)).reshape(-1), keys.split(k, 12))
    p = keys.permutation(keys.fold_in(k, 2), 4096)
    assert torch.equal(torch.sort(p).values, torch.arange(4096))
    assert not torch.equal(p, keys.permutation(keys.fold_in(k, 3), 4096))
    r = keys.randint(keys.split(k, 10000), (), 0, 500)
    assert int(r.min()) >= 0 and int(r.max()) < 500 and r.unique().numel() > 400
    u = keys.uniform(keys.split(k, 1000), (4,))
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0


def test_minibatch_indices_partition_envs():
    inds = minibatch_indices(keys.key(1), 64, 3, 4)
    assert inds.shape == (12, 16) and inds.dtype == torch.int64
    for e in range(3):
        assert torch.equal(torch.sort(inds[4 * e:4 * e + 4].reshape(-1)).values, torch.arange(64))
    assert not torch.equal(inds[:4], inds[4:8])


def test_should_run_and_config_defaults():
    # ppo.py:34-38
    assert not _should_run(100, 0, 0) and not _should_run(100, 0, -5)
    assert _should_run(0, -50, 50) and not _should_run(49, 0, 50) and _should_run(50, 0, 50)
    c = cfg.TrainConfig()
    assert (c.ppo.n_envs, c.ppo.rollout_length, c.ppo.total_steps) == (256, 20, 512_000)
    assert (c.ppo.gae_lambda, c.ppo.discounting_factor, c.ppo.clip_range) == (0.95, 0.99, 0.2)
    assert (c.ppo.n_epochs, c.ppo.n_minibatches, c.ppo.learning_rate) == (4, 4, 1e-4)
    assert c.ppo.normalize_advantages and not c.ppo.combine_advantages
    assert c.ppo.gradient_clipping is None and c.ppo.weight_decay is None
    assert c.ppo.logging_level == LoggingLevel.LOSSES and c.seed == 17
    assert c.eval.enabled and c.eval.every_steps == 50_000 and c.eval.n_envs == 64
    assert c.eval.logging_percentiles == (0, 25, 50, 75, 100)
    assert not c.video.enabled and c.checkpoint_every_steps == 500_000
    assert LoggingLevel.THROUGHPUT in LoggingLevel.ALL and LoggingLevel.BASIC == LoggingLevel.LOSSES
    assert dataclasses.replace(c, seed=3).seed == 3


def test_episode_wrapper_counters_and_flags():
    """episode_wrapper.py:12-31 / episode_wrapper_test.py:31-57."""
    env = EpisodeWrapper(MockEnv(3, 1, max_steps=1000), 10)
    s = env.reset(keys.split(keys.key(0), 32))
    assert s.info["step_counter"].dtype == torch.int64
    assert int(s.info["step_counter"].max()) < 5 and not s.info["truncated"].any()
    seen_trunc = False
    for _ in range(12):
        prev = s.info["step_counter"]
        s = env.step(s, torch.zeros(32, 1))
        assert torch.equal(s.info["step_counter"], prev + 1)
        assert torch.equal(s.info["truncated"], s.info["step_counter"] >= 10)
        assert torch.equal(s.done != 0, s.info["truncated"])  # inner env never terminates
        assert s.done.dtype == torch.float32
        seen_trunc |= bool(s.info["truncated"].any())
    assert seen_trunc and env.observation_size == 3 and env.action_size == 1


def test_mock_env_is_reproducible_and_pytree_obs():
    e1, e2 = MockEnv(5, 1, max_steps=3), MockEnv(5, 1, max_steps=3)
    k = keys.split(keys.key(9), 16)
    a, b = e1.reset(k), e2.reset(k)
    for _ in range(4):
        a, b = e1.step(a, None), e2.step(b, None)
        assert torch.equal(a.obs, b.obs)
    assert a.done.all() and float(a.reward.sum()) == 16
    c = cheetah_shaped().reset(k)
    assert set(c.obs) == {"position", "velocity"} and c.obs["velocity"].shape == (16, 9)
    d = DummyCounterEnv().reset(k)
    assert int(d.data["reset_step"].min()) >= 3 and int(d.data["reset_step"].max()) <= 9


def test_trunk_input_row_index_by_reciprocal_is_exact():
    """`csrc/mlp_bf16.hip` (input stage) maps flat element e of a [rows x K0] tile to its
    row with `(int)((e + 0.5f) * (1.0f / K0))` instead of an integer division.  Host
    restatement in IEEE fp32: exact for every K0 the kernels accept (1..512) and every
    element of the largest tile (64 rows)."""
    import numpy as np

    for K0 in range(1, 513):
        e = np.arange(64 * K0, dtype=np.int64)
        rcp = np.float32(1.0) / np.float32(K0)
        row = ((e.astype(np.float32) + np.float32(0.5)) * rcp).astype(np.int32)
        assert np.array_equal(row, (e // K0).astype(np.int32)), K0


def test_distillation_config_defaults_match_the_reference():
    """config.py:71-95 of the reference: every field and default."""
    from nnx_ppo_amd.algorithms.config import DistillationConfig, DistillationTrainConfig
    from nnx_ppo_amd.algorithms.types import LoggingLevel

    d = DistillationConfig()
    assert (d.n_envs, d.rollout_length, d.total_steps, d.learning_rate, d.n_epochs,
            d.n_minibatches, d.gradient_clipping, d.weight_decay, d.logging_level,
            d.logging_percentiles) == (256, 20, 512_000, 1e-4, 4, 4, None, None,
                                       LoggingLevel.LOSSES, None)
    t = DistillationTrainConfig()
    assert t.seed == 17 and t.checkpoint_every_steps == 500_000 and t.eval.enabled
