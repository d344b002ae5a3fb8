"""`train_ppo` / `train_distillation` run the iteration as a replayed HIP graph — the
counterpart of `nnx.jit(ppo_step)` (`nnx_ppo/algorithms/ppo.py:105,192-214`) — and must
give exactly what the eager loop gives: same callback cadence, same metrics, bit-identical
final state.  Also: the compute dtype reaches the kernels through the public keyword, and
a capture that cannot be made raises cleanly (and leaves the process usable)."""
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _setup(seed=3):
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    env = EpisodeWrapper(MockEnv(5, 1, max_steps=6), 50)
    net = factories.make_mlp_actor_critic(5, 1, [64, 64], [128, 128], Rngs(seed))
    return env, net


def _cfg(iters=6, level=None, **kw):
    from nnx_ppo_amd.algorithms.config import EvalConfig, PPOConfig, TrainConfig
    from nnx_ppo_amd.algorithms.types import LoggingLevel

    return TrainConfig(
        ppo=PPOConfig(n_envs=64, rollout_length=8, total_steps=64 * 8 * iters, n_epochs=2,
                      n_minibatches=2, learning_rate=1e-3,
                      logging_level=level or (LoggingLevel.LOSSES | LoggingLevel.GRAD_NORM)),
        eval=EvalConfig(enabled=True, every_steps=64 * 8 * 3, n_envs=8, max_episode_length=10),
        checkpoint_every_steps=64 * 8 * 2, seed=11, **kw)


def _run(dev, **kw):
    from nnx_ppo_amd.algorithms import ppo

    env, net = _setup()
    logs, ckpts = [], []
    res = ppo.train_ppo(env, net, _cfg(), log_fn=lambda m, s: logs.append((s, dict(m))),
                        checkpoint_fn=lambda ts, s: ckpts.append(
                            (s, int(ts.steps_taken), ts.optimizer.params.clone())), **kw)
    return res, net, logs, ckpts


def test_train_ppo_dw_slabs_in_adam_equal_separate_reduction(dev, monkeypatch):
    """Without clipping / GRAD_NORM the dW slab reduction rides on the Adam launch
    (`Optimizer.begin(defer_dw=True)`); the run must be bit-identical to the one that
    reduces the slabs in their own launch (MIPPO_DEFER_DW=0)."""
    from nnx_ppo_amd import ops, optim
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.types import LoggingLevel

    out = []
    for defer in (True, False):
        monkeypatch.setattr(optim, "DEFER_DW", defer)
        env, net = _setup()
        logs = []
        res = ppo.train_ppo(env, net, _cfg(level=LoggingLevel.LOSSES), compute_dtype="bf16",
                            log_fn=lambda m, s: logs.append(dict(m)), hip_graph=True)
        out.append((res.training_state.optimizer.params.clone(),
                    res.training_state.optimizer.m.clone(), logs))
        assert ops.slab_defer.pending is None and ops.slab_defer.arena is None
        # which entries an eager iteration goes through
        from nnx_ppo_amd import _lib, config
        ts = res.training_state
        with config.use_compute_dtype("bf16"), _lib.profiler as prof:
            ppo.ppo_step(env, ts, 64, 8, 0.95, 0.99, 0.2, True, False, 2, 2)
        used = {name for name, *_ in prof.records}
        assert ("mi_adam_step_slabs_f32" in used) == defer, used
        assert ("mi_dense_bwd_dw_grouped_slabs_bf16" in used) == defer, used
    (pa, ma, la), (pb, mb, lb) = out
    assert torch.equal(pa, pb) and torch.equal(ma, mb)
    for x, y in zip(la, lb):
        for k in x:
            if not k.startswith("throughput/"):
                assert float(x[k]) == float(y[k]), k


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_train_ppo_graph_equals_eager(dev, compute):
    """Iteration 1 eager, 2 recorded, 3+ replayed == every iteration launched from Python:
    bit-identical parameters, metrics and checkpointed states; same callback cadence."""
    ra, na, la, ca = _run(dev, hip_graph=True, compute_dtype=compute)
    rb, nb, lb, cb = _run(dev, hip_graph=False, compute_dtype=compute)
    assert ra.total_iterations == rb.total_iterations == 6
    assert ra.total_steps == rb.total_steps == 64 * 8 * 6
    for p, q in zip(na.parameters(), nb.parameters()):
        assert torch.equal(p.data, q.data)
    assert torch.equal(ra.training_state.optimizer.m, rb.training_state.optimizer.m)
    assert int(ra.training_state.optimizer.step) == int(rb.training_state.optimizer.step) == 24
    assert [s for s, _ in la] == [s for s, _ in lb] == [0] + [512 * i for i in range(1, 7)]
    for (sa, ma), (sb, mb) in zip(la, lb):
        assert set(ma) == set(mb)
        for k in ma:
            if k.startswith("throughput/"):
                continue
            assert float(ma[k]) == float(mb[k]), (sa, k, float(ma[k]), float(mb[k]))
    assert [c[:2] for c in ca] == [c[:2] for c in cb] == [(0, 0), (1024, 1024), (2048, 2048),
                                                          (3072, 3072)]
    for x, y in zip(ca, cb):  # the state handed to checkpoint_fn is the state AT that step
        assert torch.equal(x[2], y[2])
    assert [h["step"] for h in ra.eval_history] == [h["step"] for h in rb.eval_history]
    for ha, hb in zip(ra.eval_history, rb.eval_history):
        for k in ha:
            assert float(ha[k]) == float(hb[k]), k


def test_train_ppo_overlap_equals_strict(dev):
    """Enqueuing iteration i+1 before reading iteration i's metrics changes no value."""
    ra, na, la, _ = _run(dev, overlap_logging=True)
    rb, nb, lb, _ = _run(dev, overlap_logging=False)
    for p, q in zip(na.parameters(), nb.parameters()):
        assert torch.equal(p.data, q.data)
    for (sa, ma), (sb, mb) in zip(la, lb):
        assert sa == sb
        for k in ma:
            if not k.startswith("throughput/"):
                assert float(ma[k]) == float(mb[k]), (sa, k)


def test_metrics_reach_the_host_as_detached_scalars(dev):
    """`log_fn` gets 0-d CPU tensors that later iterations do not overwrite."""
    _, _, logs, _ = _run(dev)
    m1, m5 = logs[1][1], logs[5][1]
    for k, v in m1.items():
        if isinstance(v, torch.Tensor):
            assert v.device.type == "cpu" and v.dim() == 0, k
    assert int(m1["total_steps"]) == 512 and int(m5["total_steps"]) == 512 * 5
    assert float(m1["losses/critic/mean"]) != float(m5["losses/critic/mean"])


def test_compute_dtype_keyword_and_backend_config(dev):
    from nnx_ppo_amd import config as mi_config
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.config import BackendConfig

    assert mi_config.compute_dtype() == "f32"
    outs = {}
    for name, kw in (("f32", {}), ("bf16_kw", {"compute_dtype": "bf16"}),
                     ("bf16_cfg", {"backend": BackendConfig(compute_dtype="bf16")})):
        env, net = _setup()
        if "backend" in kw:
            res = ppo.train_ppo(env, net, _cfg(iters=3, backend=kw["backend"]))
        else:
            res = ppo.train_ppo(env, net, _cfg(iters=3), **kw)
        outs[name] = res.training_state.optimizer.params.clone()
        assert mi_config.compute_dtype() == "f32"  # restored
    assert torch.equal(outs["bf16_kw"], outs["bf16_cfg"])
    assert not torch.equal(outs["f32"], outs["bf16_kw"])
    assert float((outs["f32"] - outs["bf16_kw"]).abs().max()) < 5e-2
    with pytest.raises(ValueError):
        ppo.train_ppo(*_setup(), _cfg(iters=1), compute_dtype="fp8")


def test_train_distillation_graph_equals_eager(dev):
    from nnx_ppo_amd.algorithms import distillation
    from nnx_ppo_amd.algorithms.config import (DistillationConfig, DistillationTrainConfig,
                                               EvalConfig)
    from nnx_ppo_amd.envs import cartpole_shaped
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    outs = []
    for graph in (True, False):
        env = EpisodeWrapper(cartpole_shaped(max_steps=10), 10)
        teacher = factories.make_mlp_actor_critic(5, 1, [32, 32], [32], Rngs(1))
        student = factories.make_mlp_actor_critic(5, 1, [16, 16], [16], Rngs(2))
        cfg = DistillationTrainConfig(
            distillation=DistillationConfig(n_envs=32, rollout_length=4, total_steps=32 * 4 * 5,
                                            n_epochs=2, n_minibatches=2, learning_rate=1e-3),
            eval=EvalConfig(enabled=False))
        logged = []
        res = distillation.train_distillation(
            env, teacher, student, cfg, hip_graph=graph,
            log_fn=lambda m, s: logged.append((s, {k: float(v) for k, v in m.items()})))
        assert res.total_iterations == 5 and res.total_steps == 32 * 4 * 5
        outs.append((res.training_state.optimizer.params.clone(), logged))
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]


_FAILED_CAPTURE = textwrap.dedent("""
    import sys
    sys.path.insert(0, {root!r})
    import torch
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.config import EvalConfig, PPOConfig, TrainConfig
    from nnx_ppo_amd.algorithms.graph import GraphCaptureError
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    class SyncingEnv(MockEnv):
        # reads a device value on the host in every step: legal eagerly, uncapturable
        def step(self, state, action):
            out = super().step(state, action)
            self.last_reward = float(out.reward.sum().item())
            return out

    env = SyncingEnv(5, 1, max_steps=6)
    net = factories.make_mlp_actor_critic(5, 1, [32, 32], [32], Rngs(0))
    cfg = TrainConfig(ppo=PPOConfig(n_envs=32, rollout_length=4, total_steps=32 * 4 * 4,
                                    n_epochs=1, n_minibatches=2),
                      eval=EvalConfig(enabled=False))
    try:
        ppo.train_ppo(env, net, cfg)
    except GraphCaptureError as exc:
        print("RAISED", type(exc).__name__)
    else:
        print("NO ERROR")
        sys.exit(3)
    # the process is still usable: the same env / network train eagerly afterwards
    res = ppo.train_ppo(env, net, cfg, hip_graph=False)
    torch.cuda.synchronize()
    assert res.total_iterations == 4
    assert all(bool(torch.isfinite(p.data).all()) for p in net.parameters())
    # ... and a capturable env records and replays a graph in the same process
    env2 = MockEnv(5, 1, max_steps=6)
    net2 = factories.make_mlp_actor_critic(5, 1, [32, 32], [32], Rngs(0))
    res2 = ppo.train_ppo(env2, net2, cfg)
    assert res2.total_iterations == 4
    print("EAGER AND GRAPH OK AFTER FAILED CAPTURE")
""")


def test_capture_failure_is_clean(dev):
    """an uncapturable op inside `ppo_step` gives a clean
    `GraphCaptureError` (no silent eager fallback), and the process can still run eager
    iterations — and capture another graph — afterwards.  Run in a child process so that a
    crash there would be a test failure rather than the end of the test session."""
    r = subprocess.run([sys.executable, "-c", _FAILED_CAPTURE.format(root=str(ROOT))],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "RAISED GraphCaptureError" in r.stdout
    assert "EAGER AND GRAPH OK AFTER FAILED CAPTURE" in r.stdout


def test_bench_timing_definition_matches_train_ppo(dev):
    """The runner the bench times is the runner `train_ppo` uses: per-iteration metrics
    equal, launch mode reported."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.loop import IterationRunner

    env, net = _setup()
    ts = ppo.new_training_state(env, net, 64, 11, 1e-3, device=dev)
    fn = lambda st: ppo.ppo_step(env, st, 64, 8, 0.95, 0.99, 0.2, True, False, 2, 2)
    r = IterationRunner(fn, ts, hip_graph=True)
    got = []
    t = r.launch()
    for i in range(5):
        nxt = r.launch() if i < 4 else None
        got.append(r.collect(t))
        t = nxt
    assert r.launch_mode.startswith("hip-graph")
    assert int(r.state.steps_taken) == 64 * 8 * 5
    vals = [float(m["losses/critic/mean"]) for m in got]
    assert all(np.isfinite(v) for v in vals) and len(set(vals)) == 5
    with pytest.raises(RuntimeError):
        r.collect(0)  # long overwritten
