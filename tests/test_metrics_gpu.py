"""Values — not just presence — of the logged metric families and of the evaluation
rollout, against the oracle's restatement (`oracle/metrics.py`):

  eval_rollout                         nnx_ppo/algorithms/rollout.py:97-148
  GRAD_NORM, CRITIC_EXTRA, ACTOR_EXTRA nnx_ppo/algorithms/ppo.py:313-315,509-528
  mean/std and percentile reductions   nnx_ppo/algorithms/metrics.py:17-100

The reference holds no numeric vector for any of these (its tests check finiteness and
key presence), so the oracle is pinned here by closed forms where the env offers one
(MockEnv: the first done comes at a known step; DummyCounter: reward is 0/1) and the
product is compared with the oracle on the same env, keys and weights.  fp32 kernels vs
fp64 oracle: 1e-3 relative on loss-derived numbers, 1e-5 on counts."""
import numpy as np
import pytest
import torch

from nnx_ppo_amd import random as keys
from oracle import metrics as om
from oracle import networks as on
from oracle import ppo as op

pytestmark = pytest.mark.gpu


def _make(obs, act, ah, ch, seed=17):
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    return factories.make_mlp_actor_critic(obs, act, ah, ch, Rngs(seed))


@pytest.mark.parametrize("pct", [None, (0, 25, 50, 75, 100)])
@pytest.mark.parametrize("env_name", ["mock_wrapped", "move_to_center", "dummy_counter"])
def test_eval_rollout_vs_oracle(dev, env_name, pct):
    from nnx_ppo_amd.algorithms import rollout
    from nnx_ppo_amd.envs import DummyCounterEnv, MockEnv, MoveToCenterEnv
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    n_envs, length = 48, 14
    if env_name == "mock_wrapped":
        mk = lambda: EpisodeWrapper(MockEnv(5, 1, max_steps=9), 12)
        net = _make(5, 1, [32, 32], [32])
    elif env_name == "move_to_center":
        mk = lambda: MoveToCenterEnv(reward_falloff=1.0, border_radius=1.5)
        net = _make(2, 2, [32, 32], [32])
    else:
        mk = lambda: DummyCounterEnv()
        net = _make(1, 1, [16], [16])
    net.to(dev)
    onet = on.from_product(net)
    net.eval()
    onet.eval()
    got = rollout.eval_rollout(mk(), net, n_envs, length, keys.key(5, dev), pct)
    want = om.eval_rollout(mk(), onet, n_envs, length, keys.key(5), keys, pct)
    net.train()
    assert set(got) == set(want), (sorted(got), sorted(want))
    for k in want:
        assert np.allclose(float(got[k]), want[k], rtol=1e-4, atol=1e-5), (k, float(got[k]), want[k])
    if pct is None:
        assert {"lifespan_mean", "lifespan_std", "episode_reward/mean",
                "episode_reward/std"} == set(got)
    else:
        assert {f"lifespan/p{p}" for p in pct} <= set(got)
        assert {f"episode_reward/p{p}" for p in pct} <= set(got)
        assert "episode_reward/mean" not in got
    # closed forms (sticky done: reward stops after the first done, lifespan counts the
    # steps before it)
    if env_name == "mock_wrapped" and pct is None:
        # reward 1.0 per step up to AND including the step that ends the episode; the
        # wrapper starts each env at a random counter in [0, 6), ends it at 12 at the
        # latest, the inner env at 9 steps: lifespan = min(9, 12 - c0) - 1, reward = +1
        assert np.isclose(float(got["episode_reward/mean"]), float(got["lifespan_mean"]) + 1.0)
        assert 5.0 <= float(got["lifespan_mean"]) <= 8.0
    if env_name == "dummy_counter" and pct is None:
        assert 0.0 <= float(got["episode_reward/mean"]) <= 9.0
        assert 2.0 <= float(got["lifespan_mean"]) <= 8.0  # resets after 3..9 steps


def test_eval_rollout_sticky_done_hand_case(dev):
    """A hand-checkable case: MockEnv(max_steps=4), no wrapper.  Every env is done at its
    4th step; the episode is 10 steps long.  lifespan = 3 (steps 1-3 leave the env
    alive), reward = 4 (the step that ends the episode still pays), zero spread."""
    from nnx_ppo_amd.algorithms import rollout
    from nnx_ppo_amd.envs import MockEnv

    net = _make(5, 1, [16], [16]).to(dev)
    net.eval()
    m = rollout.eval_rollout(MockEnv(5, 1, max_steps=4), net, 16, 10, keys.key(0, dev),
                             (0, 50, 100))
    for p in (0, 50, 100):
        assert float(m[f"lifespan/p{p}"]) == 3.0 and float(m[f"episode_reward/p{p}"]) == 4.0
    assert float(m["lifespan_mean"]) == 3.0 and float(m["lifespan_std"]) == 0.0


@pytest.mark.parametrize("pct", [None, (0, 50, 100)])
def test_logging_families_vs_oracle(dev, pct):
    """One full ppo_step with every logging family on; every logged number that is a
    function of the iteration is compared with the oracle's."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.types import LoggingLevel
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    N, T, E, MB = 64, 10, 2, 2
    mk = lambda: EpisodeWrapper(MockEnv(5, 1, max_steps=5), 40)
    net = _make(5, 1, [64, 64], [128, 128])
    env, oenv = mk(), mk()
    ts = ppo.new_training_state(env, net, N, 18, 1e-3, device=dev)
    onet = on.from_product(net)
    ots = op.new_training_state(oenv, onet, N, 18, keys, 1e-3)
    level = LoggingLevel.ALL & ~LoggingLevel.THROUGHPUT
    w0 = torch.cat([p.data.reshape(-1) for p in net.parameters()]).cpu().double().numpy()
    ts, m = ppo.ppo_step(env, ts, N, T, 0.95, 0.99, 0.2, True, False, E, MB, 1.0, level, pct)
    ots, info = op.ppo_step(oenv, ots, N, T, 0.95, 0.99, 0.2, True, E, MB, keys)
    ro = info["rollout"]
    want: dict = {}
    for name in ("actor", "critic", "regularization", "clipping_fraction"):
        om.log_metric(want, f"losses/{name}", info[name], pct)
    om.log_metric(want, "grad_norm", info["grad_norm"], pct)
    om.log_metric(want, "losses/advantages", info["advantages"], pct)
    om.log_metric(want, "losses/critic_R^2", info["critic_R^2"], pct)
    om.log_metric(want, "loglikelihood", ro.loglikelihoods, pct)
    om.log_metric(want, "losses/predicted_value", ro.value_estimates, pct)
    om.log_metric(want, "rollout_batch/reward", ro.rewards, pct)
    om.log_metric(want, "rollout_batch/action", ro.actions, pct)
    want["rollout_batch/done_rate"] = float(ro.done.double().mean())
    want["rollout_batch/truncation_rate"] = float(ro.truncated.double().mean())
    missing = set(want) - set(m)
    assert not missing, missing
    loose = ("losses/actor", "losses/clipping_fraction")
    for k, w in want.items():
        g = float(m[k])
        if k.startswith(loose):
            # the actor term is a difference of O(1) numbers of size 1e-4..1e-2 here, and the
            # clipping fraction counts |r - 1| > eps events that sit near the threshold
            ok = np.isclose(g, w, rtol=2e-2, atol=2e-3)
        elif k.endswith("/std") or "/p" in k.rsplit("/", 1)[-1]:
            ok = np.isclose(g, w, rtol=5e-3, atol=2e-4)
        else:
            ok = np.isclose(g, w, rtol=1e-3, atol=2e-5)
        assert ok, (k, g, w)
    # WEIGHTS is logged over the parameters AFTER the update (ppo.py:334-335)
    w1 = torch.cat([p.data.reshape(-1) for p in net.parameters()]).cpu().double().numpy()
    ww: dict = {}
    om.log_metric(ww, "weights", torch.from_numpy(w1), pct)
    for k, w in ww.items():
        assert np.isclose(float(m[k]), w, rtol=1e-4, atol=1e-6), k
    assert not np.array_equal(w0, w1)
    # exact identities
    assert float(m["rollout_batch/done_rate"]) == float(np.float32(want["rollout_batch/done_rate"]))
    assert int(m["total_steps"]) == N * T
    if pct is None:
        assert abs(float(m["losses/advantages/mean"])) < 1e-5  # normalised per minibatch
        assert abs(float(m["losses/advantages/std"]) - 1.0) < 1e-3


def test_percentiles_helper_matches_numpy(dev):
    from nnx_ppo_amd.algorithms.metrics import percentiles

    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 1000, 30 * 4096):
        x = rng.normal(size=n).astype(np.float32)
        lv = (0, 1, 25, 50, 75, 99, 100)
        got = percentiles(torch.from_numpy(x).to(dev), lv).cpu().numpy()
        assert np.allclose(got, np.percentile(x.astype(np.float64), lv), rtol=1e-5, atol=1e-6), n


def test_col_mean_std_kernel(dev):
    """mi_col_mean_std_f32 == numpy mean / population std of every column, with a scale."""
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(0)
    for R, C in ((1, 1), (16, 4), (5, 70), (64, 3)):
        x = rng.normal(3.0, 2.0, size=(R, C)).astype(np.float32)
        out = ops.col_mean_std(torch.from_numpy(x).to(dev), 0.5).cpu().numpy()
        xs = (x * np.float32(0.5)).astype(np.float64)
        assert np.allclose(out[0], xs.mean(0), rtol=1e-6, atol=1e-7)
        assert np.allclose(out[1], xs.std(0), rtol=1e-5, atol=1e-6)
