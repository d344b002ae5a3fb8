"""End-to-end parity of the HIP PPO path against the CPU oracle, plus the
reference's own integration-test identities run on the
product."""
import numpy as np
import pytest
import torch

from nnx_ppo_amd import random as keys
from oracle import keys as okeys  # the oracle's own key scheme (numpy; tests/test_oracle_keys.py)
from oracle import networks as on
from oracle import ppo as op

pytestmark = pytest.mark.gpu
D = torch.float64


def _make(obs_size, act, actor_h, critic_h, seed=17, activation="relu", normalize=True,
          entropy_weight=1e-2):
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    return factories.make_mlp_actor_critic(obs_size, act, actor_h, critic_h, Rngs(seed),
                                           activation=activation, normalize_obs=normalize,
                                           entropy_weight=entropy_weight)


def _cpu(t):
    return t.detach().cpu()


@pytest.mark.parametrize("activation", ["relu", "tanh", "swish"])
def test_ppo_loss_gradients_vs_oracle_autograd(dev, activation):
    """One loss evaluation on one minibatch: loss terms, GAE, and EVERY parameter
    gradient against fp64 autograd through the reference-shaped T-step scan."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.types import LoggingLevel, Transition
    from nnx_ppo_amd.networks.types import PPONetworkOutput
    from nnx_ppo_amd.optim import Optimizer

    T, B, O, A = 9, 48, 5, 2
    net = _make(O, A, [32, 32], [64, 64], activation=activation)
    net.to(dev)
    opt = Optimizer(net, 1e-4, device=dev)
    rng = np.random.default_rng(0)
    # non-trivial normaliser statistics
    net.update_statistics([torch.tensor(rng.normal(1, 2, size=(4, 8, O)), dtype=torch.float32,
                                        device=dev), {"action": None, "value": None}]
                          if False else _extras_for(net, rng, O, dev))
    onet = on.from_product(net)
    obs = rng.normal(size=(T, B, O)).astype(np.float32)
    nobs = rng.normal(size=(T, B, O)).astype(np.float32)
    raw = rng.normal(size=(T, B, A)).astype(np.float32)
    ll_old = rng.normal(-1, 0.3, size=(T, B)).astype(np.float32)
    rew = rng.normal(size=(T, B)).astype(np.float32)
    done = rng.random((T, B)) < 0.15
    trunc = done & (rng.random((T, B)) < 0.5)
    g = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt).to(dev)
    extras = [g(obs), {"action": [None] * (len(net.layers[1].action.layers) - 1) + [g(raw)],
                       "value": [None] * len(net.layers[1].value.layers)}]
    mb = Transition(obs=g(obs), network_output=PPONetworkOutput(None, g(ll_old), None),
                    rewards=g(rew), done=g(done, torch.bool), truncated=g(trunc, torch.bool),
                    next_obs=g(nobs[-1:]), metrics={}, rollout_extras=extras)
    state = net.initialize_state(B)
    opt.begin()
    loss_out = torch.zeros(4, device=dev)
    ppo.ppo_loss(net, state, mb, 0.2, True, False, 0.99, 0.95, 0.5, LoggingLevel.LOSSES,
                 loss_out=loss_out)
    # oracle
    c = lambda a, dt=D: torch.as_tensor(a, dtype=dt)
    oextras = [c(obs), {"action": [None] * (len(net.layers[1].action.layers) - 1) + [c(raw)],
                        "value": [None] * len(net.layers[1].value.layers)}]
    omb = op.Transition(obs=c(obs, torch.float32), loglikelihoods=c(ll_old), rewards=c(rew),
                        done=c(done, torch.bool), truncated=c(trunc, torch.bool),
                        next_obs=c(nobs, torch.float32), rollout_extras=oextras)
    total, lm = op.ppo_loss(onet, onet.initialize_state(B), omb, 0.2, True, 0.99, 0.95, 0.5)
    params = onet.parameters()
    grads = torch.autograd.grad(total, params)
    got = _cpu(loss_out).numpy()
    assert np.allclose(got[0], lm["actor"].item(), rtol=1e-4, atol=1e-6)
    assert np.allclose(got[1], lm["critic"].item(), rtol=1e-4, atol=1e-6)
    assert np.allclose(got[2], lm["regularization"].item(), rtol=1e-4, atol=1e-6)
    assert abs(got[3] - lm["clipping_fraction"].item()) < 1e-3
    pg = [p.grad for p in net.parameters()]
    assert len(pg) == len(grads)
    for (name, p), want in zip(net.named_parameters(), grads):
        w = want.numpy()
        assert np.allclose(_cpu(p.grad).numpy(), w, rtol=2e-3, atol=1e-5 * max(1.0, np.abs(w).max())), name


def _extras_for(net, rng, O, dev):
    x = torch.tensor(rng.normal(1, 2, size=(4, 8, O)), dtype=torch.float32, device=dev)
    ad = net.layers[1]
    return [x, {"action": [None] * len(ad.action.layers), "value": [None] * len(ad.value.layers)}]


def _run_both(dev, env_fn, net_fn, N, T, n_epochs, n_mb, iters, oenv_fn=None, **kw):
    """`oenv_fn`: the oracle's own env (oracle/envs.py) where one exists; otherwise the
    product's env class on CPU tensors."""
    from nnx_ppo_amd.algorithms import ppo

    env = env_fn()
    net = net_fn()
    ts = ppo.new_training_state(env, net, N, 18, 1e-3, kw.get("clip"), kw.get("wd"), device=dev)
    onet = on.from_product(net)
    oenv = (oenv_fn or env_fn)()
    ots = op.new_training_state(oenv, onet, N, 18, okeys, 1e-3, kw.get("clip"), kw.get("wd"))
    out = []
    for _ in range(iters):
        ts, m = ppo.ppo_step(env, ts, N, T, 0.95, 0.99, 0.2, True, False, n_epochs, n_mb)
        ots, info = op.ppo_step(oenv, ots, N, T, 0.95, 0.99, 0.2, True, n_epochs, n_mb, okeys)
        out.append((ts, m, ots, info))
    return net, onet, out


def test_ppo_step_trace_vs_oracle(dev):
    """Full iteration(s): same env, same keys, same weights.  Discrete events
    (obs stream, done/trunc flags, step counters, minibatch indices, reset keys)
    bit-exact; losses per gradient step within 1e-3 rel of the fp64 oracle;
    normaliser statistics within 1e-5."""
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    N, T = 64, 12
    from oracle import envs as oe

    env_fn = lambda: EpisodeWrapper(MockEnv(5, 1, max_steps=5), 40)
    # the oracle side runs on the oracle's own env, wrapper and keys (oracle/envs.py, keys.py)
    oenv_fn = lambda: oe.EpisodeWrapper(oe.MockEnv(5, 1, max_steps=5), 40)
    net_fn = lambda: _make(5, 1, [64, 64, 64, 64], [256, 256])
    net, onet, out = _run_both(dev, env_fn, net_fn, N, T, 2, 4, iters=2, oenv_fn=oenv_fn)
    for k, (ts, m, ots, info) in enumerate(out):
        assert int(ts.steps_taken) == (k + 1) * N * T == ots.steps_taken
        assert torch.equal(_cpu(ts.rng_key), ots.rng_key)
        assert torch.equal(_cpu(ts.env_states.info["step_counter"]),
                           ots.env_states.info["step_counter"])
        assert torch.equal(_cpu(ts.env_states.obs), ots.env_states.obs)
        for name, row in (("actor", "losses/actor"), ("critic", "losses/critic"),
                          ("regularization", "losses/regularization")):
            want = info[name].numpy()
            assert np.allclose(m[row + "/mean"].item(), want.mean(), rtol=1e-3, atol=1e-5), (k, name)
            assert np.allclose(m[row + "/std"].item(), want.std(), rtol=2e-2, atol=1e-5), (k, name)
    norm, onorm = net.layers[0], onet.layers[0]
    assert float(norm.counter.value.item()) == 2 * N * T == float(onorm.counter)
    assert np.allclose(_cpu(norm.mean.value).numpy(), onorm.mean.numpy(), atol=1e-5)
    assert np.allclose(_cpu(norm.M2.value).numpy(), onorm.M2.numpy(), rtol=1e-5, atol=1e-3)
    # parameters after 16 Adam steps stay close (lr 1e-3; sign-sensitive first steps allowed a
    # small tail): 99% of entries within 1e-4, all within 5e-3
    for p, q in zip(net.parameters(), onet.parameters()):
        d = np.abs(_cpu(p.data).numpy() - q.detach().numpy())
        assert d.max() < 5e-3 and np.quantile(d, 0.99) < 2e-4


def test_rollout_bit_exact_events_and_shapes(dev):
    """rollout_test.py:31-119 (leaf shapes) and exact agreement of the event
    stream with the oracle's rollout."""
    from nnx_ppo_amd.algorithms import rollout
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    N, T = 256, 30
    env = EpisodeWrapper(MockEnv(5, 1, max_steps=13), 20)
    net = _make(5, 1, [16, 16], [16, 16])
    net.to(dev)
    onet = on.from_product(net)  # twin BEFORE the product consumes noise offsets
    k0, k1 = keys.key(3, dev), keys.key(4, dev)
    es = env.reset(keys.split(k0, N))
    ns, es2, ro = rollout.unroll_env(env, es, net, net.initialize_state(N), T, k1)
    assert ro.obs.shape == (T, N, 5) and ro.next_obs.shape == (T, N, 5)
    assert ro.network_output.actions.shape == (T, N, 1)
    assert ro.network_output.loglikelihoods.shape == (T, N)
    assert ro.network_output.value_estimates.shape == (T, N) == ro.rewards.shape
    assert ro.done.dtype == torch.bool and ro.truncated.dtype == torch.bool
    assert ro.rollout_extras[0].shape == (T, N, 5)  # normaliser input history
    assert ro.rollout_extras[1]["action"][-1].shape == (T, N, 1)  # raw actions
    oenv = EpisodeWrapper(MockEnv(5, 1, max_steps=13), 20)
    oes = oenv.reset(keys.split(keys.key(3), N))
    _, oes2, oro = op.unroll_env(oenv, oes, onet, onet.initialize_state(N), T,
                                 keys.split(keys.key(4), (T, N)))
    assert torch.equal(_cpu(ro.obs), oro.obs)
    assert torch.equal(_cpu(ro.done), oro.done) and torch.equal(_cpu(ro.truncated), oro.truncated)
    assert int(ro.done.sum()) >= N and int(ro.truncated.sum()) > 0
    assert int((ro.done & ~ro.truncated).sum()) > 0
    assert torch.equal(_cpu(es2.info["step_counter"]), oes2.info["step_counter"])
    assert int(es2.info["step_counter"].max()) <= 20  # episode_wrapper_test.py:31-57
    assert np.allclose(_cpu(ro.network_output.actions).numpy(), oro.actions.numpy(), atol=2e-5)
    assert np.allclose(_cpu(ro.network_output.loglikelihoods).numpy(), oro.loglikelihoods.numpy(),
                       atol=2e-4)
    assert np.allclose(_cpu(ro.network_output.value_estimates).numpy(),
                       oro.value_estimates.numpy(), atol=2e-5)


def test_dummy_counter_lock_step(dev):
    """rollout_test.py:121-192: sum(rewards) == T*N and 2N <= sum(done) < 10N —
    the carry reset is in lock-step with the env reset."""
    from dummies import DummyCounterNet, RepeatAndCountNet
    from nnx_ppo_amd.algorithms import rollout
    from nnx_ppo_amd.envs import DummyCounterEnv, MockEnv

    N, T = 256, 30
    env = DummyCounterEnv()
    net = DummyCounterNet().to(dev)
    es = env.reset(keys.split(keys.key(0, dev), N))
    _, _, ro = rollout.unroll_env(env, es, net, net.initialize_state(N), T, keys.key(1, dev))
    assert float(ro.rewards.sum().item()) == T * N
    nd = int(ro.done.sum().item())
    assert 2 * N <= nd < 10 * N
    # rollout_test.py:194-222: the network is called exactly T*N sample-times
    net2 = RepeatAndCountNet().to(dev)
    env2 = MockEnv(3, 3, max_steps=4)
    es = env2.reset(keys.split(keys.key(0, dev), 16))
    rollout.unroll_env(env2, es, net2, net2.initialize_state(16), 9, keys.key(1, dev))
    assert net2.n_calls == 16 * 9


def test_train_ppo_end_to_end_and_callbacks(dev):
    """ppo_test.py:220-227 (train_ppo reaches total_steps), ppo_test.py:340-349
    (normaliser counter), checkpointing_test.py:429-498 (callback cadence:
    fires at step 0 and every checkpoint_every_steps)."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.config import EvalConfig, PPOConfig, TrainConfig
    from nnx_ppo_amd.algorithms.types import LoggingLevel
    from nnx_ppo_amd.envs import MoveToCenterEnv

    env = MoveToCenterEnv(reward_falloff=1.0, border_radius=10.0)
    net = _make(2, 2, [32, 32], [32, 32])
    cfg = TrainConfig(
        ppo=PPOConfig(n_envs=32, rollout_length=8, total_steps=32 * 8 * 5, n_epochs=2,
                      n_minibatches=2, gradient_clipping=1.0, weight_decay=True,
                      logging_level=LoggingLevel.ALL),
        eval=EvalConfig(enabled=True, every_steps=32 * 8 * 2, n_envs=8, max_episode_length=20),
        checkpoint_every_steps=32 * 8 * 2)
    logs, ckpts = [], []
    res = ppo.train_ppo(env, net, cfg, seed=5, log_fn=lambda m, s: logs.append((s, dict(m))),
                        checkpoint_fn=lambda ts, s: ckpts.append(s))
    assert res.total_steps == 32 * 8 * 5 and res.total_iterations == 5
    assert ckpts == [0, 512, 1024] and ckpts[0] == 0
    assert [s for s, _ in logs] == [0, 256, 512, 768, 1024, 1280]
    assert len(res.eval_history) == 3 and res.eval_history[0]["step"] == 0
    for s, m in logs[1:]:
        for k in ("losses/actor/mean", "losses/critic/mean", "losses/regularization/mean",
                  "losses/actor/std", "total_steps", "grad_norm/mean", "weights/mean",
                  "losses/clipping_fraction/mean", "throughput/train_sps",
                  # CRITIC_EXTRA / ACTOR_EXTRA / TRAIN_ROLLOUT_STATS (ppo.py:509-528,
                  # metrics.py:36-66)
                  "losses/advantages/mean", "losses/critic_R^2/mean",
                  "losses/predicted_value/mean", "loglikelihood/mean",
                  "rollout_batch/reward/mean", "rollout_batch/done_rate"):
            assert k in m, k
        # normalised advantages: zero mean over every gradient step's minibatch
        assert abs(float(m["losses/advantages/mean"])) < 1e-4
        assert float(m["losses/critic_R^2/mean"]) <= 1.0
        for k, v in m.items():
            val = v if isinstance(v, float) else float(v)
            assert np.isfinite(val), k
    assert "episode_reward/p50" in logs[0][1] and "lifespan_mean" in logs[0][1]
    assert float(net.layers[0].counter.value.item()) == 5 * 32 * 8
    for p in net.parameters():
        assert torch.isfinite(p.data).all()


def test_pytree_obs_network(dev):
    """BASELINE config 3 shape: dict observations through Normalizer({..}) ->
    Flattener -> PPOAdapter, one ppo_step, finite and counted."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import cheetah_shaped
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.containers import Sequential
    from nnx_ppo_amd.networks.normalizer import Normalizer
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.networks.utils import Flattener

    env = cheetah_shaped(max_steps=6)
    ad = factories.make_mlp_actor_critic(17, 6, [32, 32], [64, 64], Rngs(1), normalize_obs=False)
    net = Sequential([Normalizer({"position": 8, "velocity": 9}), Flattener(), ad])
    ts = ppo.new_training_state(env, net, 64, 3, device=dev)
    onet = on.from_product(net)
    oenv = cheetah_shaped(max_steps=6)
    ots = op.new_training_state(oenv, onet, 64, 3, okeys)
    ts, m = ppo.ppo_step(env, ts, 64, 10, 0.95, 0.99, 0.2, True, False, 2, 2)
    ots, info = op.ppo_step(oenv, ots, 64, 10, 0.95, 0.99, 0.2, True, 2, 2, okeys)
    assert int(ts.steps_taken) == 640
    assert np.allclose(m["losses/critic/mean"].item(), info["critic"].numpy().mean(), rtol=1e-3)
    assert np.allclose(m["losses/actor/mean"].item(), info["actor"].numpy().mean(), rtol=1e-3,
                       atol=1e-5)
    n, o = net.layers[0], onet.layers[0]
    assert float(n.counter.value.item()) == 640
    for k in ("position", "velocity"):
        assert np.allclose(_cpu(n.mean.value[k]).numpy(), o.mean[k].numpy(), atol=1e-5)


def test_graphed_step_equals_eager(dev):
    """The captured HIP graph of an iteration replays to exactly what eager
    launches compute: same kernels, same order, same device-resident RNG /
    optimiser state -> bitwise-equal parameters, statistics and metrics."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.graph import GraphedPPOStep
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    N, T = 128, 8
    args = (N, T, 0.95, 0.99, 0.2, True, False, 2, 2)

    def setup():
        env = EpisodeWrapper(MockEnv(5, 1, max_steps=5), 30)
        net = _make(5, 1, [32, 32], [64, 64])
        return env, net, ppo.new_training_state(env, net, N, 11, 1e-3, 1.0, device=dev)

    env_a, net_a, ts_a = setup()
    for _ in range(4):
        ts_a, m_a = ppo.ppo_step(env_a, ts_a, *args)
    env_b, net_b, ts_b = setup()
    g = GraphedPPOStep(env_b, ts_b, *args, warmup=1)
    for _ in range(3):
        ts_b, m_b = g()
    torch.cuda.synchronize()
    assert int(ts_a.steps_taken) == int(ts_b.steps_taken) == 4 * N * T
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):
        assert torch.equal(pa.data, pb.data)
    assert torch.equal(net_a.layers[0].mean.value, net_b.layers[0].mean.value)
    assert torch.equal(net_a.layers[0].counter.value, net_b.layers[0].counter.value)
    assert torch.equal(ts_a.env_states.obs, ts_b.env_states.obs)
    assert torch.equal(ts_a.rng_key, ts_b.rng_key)
    for k in m_a:
        assert torch.equal(torch.as_tensor(m_a[k]), torch.as_tensor(m_b[k])), k


def test_segmented_capture_equals_eager(dev, monkeypatch):
    """A sharded iteration recorded as graph / collective / graph / ... replays to the
    same bits as eager launches.  The collectives are stand-ins of a one-rank group
    (identity all-reduce, copy all-gather) so the segmentation itself is what is
    tested: 16 gradient steps x 2 + normaliser merge + loss rows = 34 cuts."""
    from nnx_ppo_amd import parallel
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.graph import SegmentedPPOStep
    from nnx_ppo_amd.envs import MockEnv

    calls = []

    class _Dist:
        class ReduceOp:
            SUM = "sum"

        @staticmethod
        def all_reduce(t, op=None):
            calls.append("ar")

        @staticmethod
        def all_gather(parts, src):
            calls.append("ag")
            parts[0].copy_(src)

        @staticmethod
        def get_world_size():
            return 1

    monkeypatch.setattr(parallel, "dist", _Dist)
    monkeypatch.setattr(parallel, "is_distributed", lambda: True)
    monkeypatch.setattr(parallel, "world_size", lambda: 1)
    monkeypatch.setattr(parallel, "rank", lambda: 0)

    def setup():
        env = MockEnv(5, 1, max_steps=7)
        net = _make(5, 1, [32, 32], [32])
        return env, net, ppo.new_training_state(env, net, 64, 9, 1e-3, device=dev)

    args = (64, 8, 0.95, 0.99, 0.2, True, False, 4, 4)
    ea, na, ta = setup()
    for _ in range(4):
        ta, ma = ppo.ppo_step(ea, ta, *args)
    calls.clear()
    eb, nb, tb = setup()
    step = SegmentedPPOStep(eb, tb, *args, warmup=1)
    n_coll = sum(1 for x in step.program if not isinstance(x, torch.cuda.CUDAGraph))
    assert n_coll == 34 and len(step.program) == 2 * n_coll + 1
    for _ in range(3):
        tb, mb = step()
    torch.cuda.synchronize()
    for pa, pb in zip(na.parameters(), nb.parameters()):
        assert torch.equal(pa.data, pb.data)
    assert int(tb.steps_taken) == int(ta.steps_taken)
    for k in ma:
        if k.startswith("losses/"):
            assert torch.equal(torch.as_tensor(ma[k]), torch.as_tensor(mb[k])), k
