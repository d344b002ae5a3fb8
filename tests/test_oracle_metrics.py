"""The oracle's `eval_rollout` / metric reductions (`oracle/metrics.py`, restating
`nnx_ppo/algorithms/rollout.py:76-148` and `metrics.py:72-100`) against hand-computed
cases.  The reference holds no numeric vector for them (its tests check key presence),
so these closed forms are what pins the restatement.  CPU only."""
import numpy as np
import torch

from nnx_ppo_amd import random as keys
from oracle import metrics as om
from oracle import networks as on


def _onet(obs, act, ah, ch):
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    return on.from_product(factories.make_mlp_actor_critic(obs, act, ah, ch, Rngs(0))).eval()


def test_eval_rollout_hand_case():
    """MockEnv(max_steps=4): done at the 4th step of every env, reward 1.0 per step.
    Sticky done => lifespan 3 (rollout.py:123 counts steps whose post-step done is
    false), cumulative reward 4 (rollout.py:118-121 masks by the PRE-step done)."""
    from nnx_ppo_amd.envs import MockEnv

    m = om.eval_rollout(MockEnv(5, 1, max_steps=4), _onet(5, 1, [8], [8]), 6, 9, keys.key(1),
                        keys, None)
    assert m == {"lifespan_mean": 3.0, "lifespan_std": 0.0, "episode_reward/mean": 4.0,
                 "episode_reward/std": 0.0}
    m = om.eval_rollout(MockEnv(5, 1, max_steps=4), _onet(5, 1, [8], [8]), 6, 9, keys.key(1),
                        keys, (0, 50, 100))
    assert set(m) == {"lifespan_mean", "lifespan_std", "episode_reward/p0",
                      "episode_reward/p50", "episode_reward/p100", "lifespan/p0",
                      "lifespan/p50", "lifespan/p100"}
    assert m["lifespan/p50"] == 3.0 and m["episode_reward/p100"] == 4.0


def test_eval_rollout_episode_shorter_than_life():
    """max_episode_length < the env's life: nothing is done, lifespan = length."""
    from nnx_ppo_amd.envs import MockEnv

    m = om.eval_rollout(MockEnv(5, 1, max_steps=50), _onet(5, 1, [8], [8]), 4, 7, keys.key(2),
                        keys, None)
    assert m["lifespan_mean"] == 7.0 and m["episode_reward/mean"] == 7.0


def test_eval_rollout_dict_rewards():
    """PyTree rewards: one metric family per key (rollout.py:83-85)."""
    from nnx_ppo_amd.envs import TwoArmEnv

    class ZeroNet(on.Module):
        def __call__(self, state, obs, extras=None):
            n = obs["arm1"]["pos"].shape[0]
            z = torch.zeros(n, 2, dtype=torch.float64)
            return on.Out((), on.PPOOut({"arm1": z, "arm2": z}, None, None), torch.zeros(()), {})

    m = om.eval_rollout(TwoArmEnv(), ZeroNet(), 5, 3, keys.key(3), keys, None)
    assert {"episode_reward/arm1/mean", "episode_reward/arm2/std"} <= set(m)
    # zero actions: the arms never move, reward exp(-|p0|) per step for 3 steps
    assert 0.0 < m["episode_reward/arm1/mean"] <= 3.0 and m["lifespan_mean"] == 3.0


def test_log_metric_rules():
    m: dict = {}
    x = torch.tensor([1.0, 2.0, 3.0, 6.0])
    om.log_metric(m, "a", x)
    assert m == {"a/mean": 3.0, "a/std": float(np.std([1, 2, 3, 6]))}
    m = {}
    om.log_metric(m, "b", torch.tensor([True, False, False, False]))
    assert m == {"b": 0.25}                      # bool arrays log their mean
    m = {}
    om.log_metric(m, "c", {"x": x, "y": {"z": x}}, (0, 50, 100))
    assert m == {"c/x/p0": 1.0, "c/x/p50": 2.5, "c/x/p100": 6.0,
                 "c/y/z/p0": 1.0, "c/y/z/p50": 2.5, "c/y/z/p100": 6.0}
