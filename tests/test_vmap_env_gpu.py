"""`envs.VmapEnv` on the GPU: a single-env env (the reference's convention) lifted with
`torch.func.vmap` drives the kernel path exactly as its hand-batched twin does —
rollout lock-step identity of `rollout_test.py:121-192`, a full `ppo_step`, and the
HIP-graph replay of `train_ppo`."""
import pytest
import torch

from nnx_ppo_amd import random as keys

pytestmark = pytest.mark.gpu


def test_lifted_dummy_counter_rollout_lock_step(dev):
    """rollout_test.py:121-192 with the env written single-env: sum(rewards) == T * N
    proves the net-carry reset stays in lock-step with the (vmapped) env reset."""
    from dummies import DummyCounterNet, SingleDummyCounterEnv
    from nnx_ppo_amd.algorithms import rollout
    from nnx_ppo_amd.envs import DummyCounterEnv, VmapEnv

    N, T = 32, 40
    outs = []
    for env in (VmapEnv(SingleDummyCounterEnv()), DummyCounterEnv()):
        net = DummyCounterNet().to(dev)
        es = env.reset(keys.split(keys.key(0, dev), N))
        _, _, ro = rollout.unroll_env(env, es, net, net.initialize_state(N), T, keys.key(1, dev))
        assert float(ro.rewards.sum()) == T * N
        assert 2 * N <= int(ro.done.sum()) < 10 * N
        outs.append(ro)
    assert torch.equal(outs[0].done, outs[1].done)
    assert torch.equal(outs[0].rewards, outs[1].rewards)


def test_ppo_step_on_a_lifted_env_equals_the_batched_twin(dev):
    from dummies import SingleMoveToCenterEnv
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import MoveToCenterEnv, VmapEnv
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    res = []
    for env in (VmapEnv(SingleMoveToCenterEnv(1.0, 2.0)), MoveToCenterEnv(1.0, 2.0)):
        net = factories.make_mlp_actor_critic(2, 2, [32, 32], [32, 32], Rngs(4))
        ts = ppo.new_training_state(env, net, 64, 7, 1e-3, device=dev)
        for _ in range(2):
            ts, m = ppo.ppo_step(env, ts, 64, 8, 0.95, 0.99, 0.2, True, False, 2, 2)
        res.append((ts.optimizer.params.clone(), ts.env_states.obs.clone(),
                    float(m["losses/critic/mean"])))
    assert torch.equal(res[0][1], res[1][1])           # same event stream
    assert torch.equal(res[0][0], res[1][0])           # same parameters, bit for bit
    assert res[0][2] == res[1][2]


def test_train_ppo_graph_on_a_lifted_env(dev):
    """The vmapped env is capturable: train_ppo replays it inside the HIP graph and gets
    what the eager loop gets."""
    from dummies import SingleMoveToCenterEnv
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.config import EvalConfig, PPOConfig, TrainConfig
    from nnx_ppo_amd.envs import VmapEnv
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    out = []
    for graph in (True, False):
        env = VmapEnv(SingleMoveToCenterEnv(1.0, 2.0))
        net = factories.make_mlp_actor_critic(2, 2, [32, 32], [32], Rngs(4))
        cfg = TrainConfig(ppo=PPOConfig(n_envs=32, rollout_length=6, total_steps=32 * 6 * 4,
                                        n_epochs=1, n_minibatches=2),
                          eval=EvalConfig(enabled=True, every_steps=32 * 6 * 2, n_envs=8,
                                          max_episode_length=10))
        r = ppo.train_ppo(env, net, cfg, hip_graph=graph)
        assert r.total_iterations == 4
        out.append(r.training_state.optimizer.params.clone())
    assert torch.equal(out[0], out[1])
