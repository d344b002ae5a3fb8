"""Distillation path (reference `nnx_ppo/algorithms/distillation.py`): the
loss and every student gradient against the CPU oracle's fp64 autograd, and the
reference's own test identities (`distillation_test.py:46-199`) run on the product."""
import numpy as np
import pytest
import torch

from oracle import distillation as od
from oracle import networks as on

pytestmark = pytest.mark.gpu
D = torch.float64


def _nets(dev, obs=5, act=2, seed=17, hidden=(16, 16)):
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    mk = lambda s: factories.make_mlp_actor_critic(obs, act, list(hidden), list(hidden), Rngs(s),
                                                   normalize_obs=True, entropy_weight=1e-2)
    teacher, student = mk(seed), mk(seed + 1)
    teacher.to(dev)
    student.to(dev)
    return teacher, student


def _cpu(t):
    return t.detach().cpu()


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_distillation_loss_and_gradients_vs_oracle(dev, compute):
    """Tolerances: fp32 path 1e-4 rel on the losses, 2e-3 rel per gradient element (as
    the PPO loss test); bf16 path 5e-2 on the losses and on each gradient tensor's
    Frobenius-relative error — the operands are rounded to bf16, the oracle is fp64."""
    from nnx_ppo_amd import config as mi_config
    from nnx_ppo_amd.algorithms import distillation
    from nnx_ppo_amd.algorithms.types import DistillationTransition, LoggingLevel
    from nnx_ppo_amd.networks.types import PPONetworkOutput
    from nnx_ppo_amd.optim import Optimizer

    prev = mi_config.compute_dtype()
    mi_config.set_compute_dtype(compute)
    try:
        T, B, O, A = 7, 40, 5, 2
        teacher, student = _nets(dev, O, A, hidden=(32, 32))
        opt = Optimizer(student, 1e-4, device=dev)
        rng = np.random.default_rng(3)
        x = torch.tensor(rng.normal(1, 2, size=(4, 8, O)), dtype=torch.float32, device=dev)
        ad = student.layers[1]
        student.update_statistics([x, {"action": [None] * len(ad.action.layers),
                                       "value": [None] * len(ad.value.layers)}])
        ostudent = on.from_product(student)
        obs = rng.normal(size=(T, B, O)).astype(np.float32)
        mu_teacher = rng.normal(scale=0.5, size=(T, B, A)).astype(np.float32)
        done = rng.random((T, B)) < 0.15
        g = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt).to(dev)
        n_a, n_v = len(ad.action.layers), len(ad.value.layers)
        extras = [g(obs), {"action": [None] * (n_a - 1) + [g(mu_teacher)], "value": [None] * n_v}]
        mb = DistillationTransition(
            obs=g(obs), student_output=PPONetworkOutput(None, None, None), rewards=None,
            done=g(done, torch.bool), truncated=None, next_obs=None, metrics={},
            student_rollout_extras=None, teacher_rollout_extras=extras)
        opt.begin()
        total, lm = distillation.distillation_loss(student, student.initialize_state(B), mb,
                                                   LoggingLevel.LOSSES)
        c = lambda a, dt=D: torch.as_tensor(a, dtype=dt)
        oextras = [c(obs), {"action": [None] * (n_a - 1) + [c(mu_teacher)],
                            "value": [None] * n_v}]
        ototal, olm = od.distillation_loss(ostudent, ostudent.initialize_state(B),
                                           c(obs, torch.float32), c(done, torch.bool), oextras)
        params = ostudent.parameters()
        grads = torch.autograd.grad(ototal, params, allow_unused=True)
        rl, rg = (1e-4, 2e-3) if compute == "f32" else (5e-2, 5e-2)
        assert np.allclose(float(lm["losses/distillation_nll"]), olm["distillation_nll"].item(),
                           rtol=rl, atol=1e-6)
        assert np.allclose(float(lm["losses/regularization"]), olm["regularization"].item(),
                           rtol=rl, atol=1e-6)
        assert np.allclose(float(total), ototal.item(), rtol=rl, atol=1e-6)
        n_checked = 0
        for (name, p), want in zip(student.named_parameters(), grads):
            got = _cpu(p.grad).numpy()
            if want is None:  # the value port gets no gradient from this loss
                assert not got.any(), name
                continue
            w = want.numpy()
            scale = max(1e-3, np.abs(w).max())
            if compute == "f32":
                assert np.allclose(got, w, rtol=rg, atol=1e-5 * scale), name
            else:  # bf16 operands: relative error of the whole tensor (Frobenius norm)
                assert np.linalg.norm(got - w) <= rg * max(np.linalg.norm(w), 1e-6), name
            n_checked += 1
        assert n_checked >= 2 * (n_a - 1)  # every actor kernel and bias (n_a counts the sampler)
    finally:
        mi_config.set_compute_dtype(prev)


def test_distillation_step_identities(dev):
    """distillation_test.py:46-150: steps counted, metrics finite, state continuous over
    two steps, the teacher's parameters untouched, the student's changed."""
    from nnx_ppo_amd.algorithms import distillation
    from nnx_ppo_amd.algorithms.types import LoggingLevel
    from nnx_ppo_amd.envs import cartpole_shaped
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    env = EpisodeWrapper(cartpole_shaped(max_steps=6), 6)
    teacher, student = _nets(dev, 5, 1)
    N, T = 8, 4
    state = distillation.new_distillation_state(env, teacher, student, N, seed=18, device=dev)
    assert int(state.steps_taken) == 0
    teacher.eval()
    t_before = [p.data.clone() for p in teacher.parameters()]
    s_before = [p.data.clone() for p in student.parameters()]
    for it in range(2):
        state, metrics = distillation.distillation_step(env, teacher, state, N, T, 2, 2,
                                                        LoggingLevel.ALL)
        assert int(state.steps_taken) == (it + 1) * N * T
        for k, v in metrics.items():
            assert bool(torch.isfinite(torch.as_tensor(v, dtype=torch.float32)).all()), k
        assert "losses/distillation_nll/mean" in metrics
    for a, p in zip(t_before, teacher.parameters()):
        assert torch.equal(a, p.data)
    assert any(not torch.equal(a, p.data) for a, p in zip(s_before, student.parameters()))


def test_teacher_extras_hold_the_teacher_mean(dev):
    """distillation.py:16-21: with the teacher in eval mode its sampler emits the mean, so
    the target stored in `teacher_rollout_extras` is `mu_teacher`."""
    teacher, _ = _nets(dev, 5, 2)
    teacher.eval()
    x = torch.randn(16, 5, device=dev)
    out = teacher(teacher.initialize_state(16), x)
    sampler_pos = len(teacher.layers[1].action.layers) - 1
    raw = out.rollout_extras[-1]["action"][sampler_pos]
    out2 = teacher(teacher.initialize_state(16), x)
    assert torch.equal(raw, out2.rollout_extras[-1]["action"][sampler_pos])  # no noise
    assert torch.allclose(torch.tanh(raw), out.output.actions, atol=1e-6)


def test_student_moves_towards_the_teacher(dev):
    """The NLL of the teacher's mean under the student falls over a few iterations."""
    from nnx_ppo_amd.algorithms import distillation
    from nnx_ppo_amd.algorithms.types import LoggingLevel
    from nnx_ppo_amd.envs import cartpole_shaped
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    env = EpisodeWrapper(cartpole_shaped(max_steps=50), 50)
    teacher, student = _nets(dev, 5, 1, hidden=(32, 32))
    teacher.eval()
    state = distillation.new_distillation_state(env, teacher, student, 256, seed=5,
                                                learning_rate=3e-3, device=dev)
    nll = []
    for _ in range(12):
        state, m = distillation.distillation_step(env, teacher, state, 256, 8, 2, 2,
                                                  LoggingLevel.LOSSES)
        nll.append(float(m["losses/distillation_nll/mean"]))
    assert nll[-1] < nll[0] - 0.05, nll


def test_train_distillation_end_to_end(dev):
    """distillation_test.py:188-199."""
    from nnx_ppo_amd.algorithms import distillation
    from nnx_ppo_amd.algorithms.config import (DistillationConfig, DistillationTrainConfig,
                                               EvalConfig)
    from nnx_ppo_amd.envs import cartpole_shaped
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    env = EpisodeWrapper(cartpole_shaped(max_steps=10), 10)
    teacher, student = _nets(dev, 5, 1)
    cfg = DistillationTrainConfig(
        distillation=DistillationConfig(n_envs=8, rollout_length=4, total_steps=96, n_epochs=2,
                                        n_minibatches=2),
        eval=EvalConfig(enabled=True, every_steps=64, n_envs=4, max_episode_length=12))
    logged = []
    res = distillation.train_distillation(env, teacher, student, cfg,
                                          log_fn=lambda m, s: logged.append(s))
    assert res.total_steps == 96 and res.total_iterations == 3
    assert len(res.eval_history) >= 2 and res.eval_history[0]["step"] == 0
    assert logged[-1] == 96
    assert "losses/distillation_nll/mean" in res.final_metrics
