"""PyTree rewards, per-key value heads, structured log-likelihoods and
`combine_advantages` (`ppo.py:440-510`) on the two-arm env of
`test_dummies/dict_obs_act_env.py:128-230`.  `test_ppo_step_combine_advantages` restates
`ppo_test.py:416-441` (finite metrics, steps_taken); the others pin losses, gradients
and parameters against the oracle's restatement of the same lines."""
import numpy as np
import pytest
import torch

from nnx_ppo_amd import random as keys
from oracle import networks as on
from oracle import ppo as op

pytestmark = pytest.mark.gpu


def _net(rngs, joint_policy=False):
    """dict obs -> flat; two value heads (one per reward key); one sampler per arm, or —
    `joint_policy` — one sampler over both arms' actions (a tensor log-likelihood)."""
    from nnx_ppo_amd.networks import activations as A
    from nnx_ppo_amd.networks.adapter import PPOAdapter
    from nnx_ppo_amd.networks.containers import Sequential, Splitter
    from nnx_ppo_amd.networks.feedforward import Dense
    from nnx_ppo_amd.networks.sampling_layers import NormalTanhSampler
    from nnx_ppo_amd.networks.utils import Flattener, Map

    smp = lambda: NormalTanhSampler(rngs, entropy_weight=1e-2, min_std=1e-1)
    if joint_policy:
        action = Sequential([Dense(8, 16, rngs, activation=A.relu), Dense(16, 8, rngs), smp()])
    else:
        action = Sequential([Dense(8, 16, rngs, activation=A.relu), Dense(16, 8, rngs),
                             Splitter(arm1=4, arm2=4), Map(arm1=smp(), arm2=smp())])
    value = Sequential([Dense(8, 16, rngs, activation=A.tanh), Dense(16, 2, rngs),
                        Splitter(arm1=1, arm2=1)])
    return Sequential([Flattener(), PPOAdapter(action=action, value=value)])


class _JointActionEnv:
    """TwoArmEnv driven by one flat `[N, 4]` action (arm1 | arm2)."""

    def __init__(self):
        from nnx_ppo_amd.envs import TwoArmEnv

        self.env = TwoArmEnv()

    def reset(self, rng):
        return self.env.reset(rng)

    def step(self, state, action):
        return self.env.step(state, {"arm1": action[..., :2], "arm2": action[..., 2:]})


def test_ppo_step_combine_advantages(dev):
    """ppo_test.py:416-441."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.types import LoggingLevel
    from nnx_ppo_amd.envs import TwoArmEnv
    from nnx_ppo_amd.networks.types import Rngs

    env, net = TwoArmEnv(), _net(Rngs(43))
    ts = ppo.new_training_state(env, net, 8, 43, device=dev)
    assert int(ts.steps_taken) == 0
    ts, metrics = ppo.ppo_step(env, ts, 8, 4, 0.95, 0.99, 0.2, True, True, 2, 2, 1.0,
                               LoggingLevel.LOSSES | LoggingLevel.CRITIC_EXTRA)
    assert int(ts.steps_taken) == 8 * 4
    for k, v in metrics.items():
        assert bool(torch.isfinite(torch.as_tensor(v, dtype=torch.float32)).all()), k


@pytest.mark.parametrize("combine,joint", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("normalize", [True, False])
def test_pytree_loss_vs_oracle(dev, combine, joint, normalize):
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import TwoArmEnv
    from nnx_ppo_amd.networks.types import Rngs

    N, T = 32, 6
    mk_env = (lambda: _JointActionEnv()) if joint else (lambda: TwoArmEnv())
    env, oenv = mk_env(), mk_env()
    net = _net(Rngs(7), joint_policy=joint)
    ts = ppo.new_training_state(env, net, N, 7, 1e-3, device=dev)
    onet = on.from_product(net)
    ots = op.new_training_state(oenv, onet, N, 7, keys, 1e-3)
    for k in range(2):
        ts, m = ppo.ppo_step(env, ts, N, T, 0.95, 0.99, 0.2, normalize, combine, 2, 2)
        ots, info = op.ppo_step(oenv, ots, N, T, 0.95, 0.99, 0.2, normalize, 2, 2, keys,
                                combine_advantages=combine)
        for name in ("actor", "critic", "regularization"):
            # per-key trees are logged as `losses/<name>/<key>/mean` (ppo.py:509-513); the
            # oracle reports their sum (`jax.tree.reduce(jp.add, ...)`, ppo.py:505-507)
            ks = [k for k in m if k.startswith(f"losses/{name}/") and k.endswith("/mean")]
            assert len(ks) == (1 if (name == "regularization" or (joint and name == "actor"))
                               else 2), ks
            got, want = sum(m[k].item() for k in ks), info[name].numpy().mean()
            assert np.isfinite(got) and np.allclose(got, want, rtol=2e-3, atol=2e-5), (k, name)
            # ... and each key's own entry against the oracle's tree of the same term
            tree = info.get(f"{name}_tree")
            if isinstance(tree, dict):
                assert {f"losses/{name}/{key}/mean" for key in tree} == set(ks), (ks, list(tree))
                for key, rows in tree.items():
                    for stat, want_k in (("mean", rows.numpy().mean()), ("std", rows.numpy().std())):
                        got_k = m[f"losses/{name}/{key}/{stat}"].item()
                        assert np.allclose(got_k, want_k, rtol=2e-3, atol=2e-5), \
                            (name, key, stat, got_k, want_k)
    for p, q in zip(net.parameters(), onet.parameters()):
        assert torch.isfinite(p.data).all()
        assert float((p.data.cpu() - q.detach()).abs().max()) < 5e-4


def test_dict_rewards_without_a_matching_policy_tree_is_an_error(dev):
    """ppo.py:494-499: `jax.tree.map` over (ll_new, ll_old, advantages) needs matching
    trees — a joint (tensor) log-likelihood with dict rewards and combine_advantages=False
    cannot be evaluated."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.networks.types import Rngs

    env = _JointActionEnv()
    net = _net(Rngs(1), joint_policy=True)
    ts = ppo.new_training_state(env, net, 8, 1, device=dev)
    with pytest.raises((ValueError, TypeError, KeyError, AssertionError)):
        ppo.ppo_step(env, ts, 8, 4, 0.95, 0.99, 0.2, True, False, 1, 1)
