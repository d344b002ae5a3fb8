"""The kernels `bench.py` times, under the oracle DIRECTLY (VERDICT r2 "weak" #1).

The benchmark's gradient step at C2 is four launches at [T, B] = [30, 1024]:
`mi_policy_ws_fwd_bf16` (both trunks, weights stationary), `mi_policy_ws_bwd_gae_bf16` (the
backward with the GAE scan, the advantage statistics and the loss gradients inside),
`mi_dense_bwd_dw_grouped_slabs_bf16` and `mi_adam_step_slabs_f32`.  The other oracle tests run
C2's network at N = 64, where B = 16 is outside those kernels' shape class and they are
reached only through bit-identity chains.  Here the whole `ppo_step`
(`nnx_ppo/algorithms/ppo.py:254-348,397-531`) runs on C2's network in bf16 at sizes that DO
dispatch to them — asserted through the C-ABI call record — against the fp64 oracle on the
oracle's own env, wrapper and keys:

  * N = 1024, T = 30, 2 minibatches -> B = 512, M = 15 360 rows per gradient step;
  * the same size replayed as a HIP graph (the in-kernel statistics hand-over and the deferred
    `mi_policy_loss_finalize_f32` are then REPLAYED, not launched from Python);
  * the full C2 size (N = 4096) through `train_ppo`, by the size-independent properties the
    reference's own tests use (`ppo_test.py:38-62,340-349`);
  * a forced time-out of the in-kernel hand-over: the iteration that sees it raises, its
    statistics are poisoned, and no checkpoint of the poisoned parameters is written
    (VERDICT r2 "weak" #2).

Bars: events bit-exact; losses at the bf16 end-to-end bars of DESIGN §5 (5e-2 actor /
regulariser, 3e-3 critic against fp64); parameters after the Adam steps within 2e-2."""
import numpy as np
import pytest
import torch

from oracle import envs as oe
from oracle import keys as okeys
from oracle import networks as on
from oracle import ppo as op

pytestmark = pytest.mark.gpu

BENCHED = {"mi_policy_ws_fwd_bf16", "mi_policy_ws_bwd_gae_bf16",
           "mi_dense_bwd_dw_grouped_slabs_bf16", "mi_adam_step_slabs_f32"}
ACTOR_H, CRITIC_H = [64, 64, 64, 64], [256, 256]   # BASELINE configs[1] (bench.py)


def _c2(seed=17, max_steps=7, max_len=40):
    from nnx_ppo_amd.envs import cartpole_shaped
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    env = EpisodeWrapper(cartpole_shaped(max_steps=max_steps), max_len)
    oenv = oe.EpisodeWrapper(oe.MockEnv(5, 1, max_steps=max_steps), max_len)
    net = factories.make_mlp_actor_critic(5, 1, ACTOR_H, CRITIC_H, Rngs(seed), normalize_obs=True)
    return env, oenv, net


def _called(prof) -> set:
    return {name for name, *_ in prof.records}


def test_c2_benched_kernels_ppo_step_vs_oracle(dev):
    from nnx_ppo_amd import _lib, config, ops
    from nnx_ppo_amd.algorithms import ppo

    N, T, E, MB = 1024, 30, 2, 2
    with config.use_compute_dtype("bf16"):
        env, oenv, net = _c2()
        ts = ppo.new_training_state(env, net, N, 23, 3e-4, device=dev)
        onet = on.from_product(net)
        ots = op.new_training_state(oenv, onet, N, 23, okeys, 3e-4)
        for k in range(2):
            with _lib.profiler as prof:
                ts, m = ppo.ppo_step(env, ts, N, T, 0.95, 0.99, 0.2, True, False, E, MB)
            ots, info = op.ppo_step(oenv, ots, N, T, 0.95, 0.99, 0.2, True, E, MB, okeys)
            used = _called(prof)
            assert BENCHED <= used, sorted(BENCHED - used)
            assert "mi_policy_loss_finalize_f32" in used   # the deferred loss sum
            # exactly the bench's four launches per gradient step: none of the forms they
            # replaced runs beside them
            assert not used & {"mi_gae_ppo_loss_f32", "mi_policy_ws_bwd_bf16",
                               "mi_policy_bwd_bf16", "mi_policy_fwd_bf16",
                               "mi_reduce_slabs_grouped_f32"}, used
            n_steps = sum(1 for name, *_ in prof.records if name == "mi_policy_ws_bwd_gae_bf16")
            assert n_steps == E * MB
            shapes = {ints for name, ints, *_ in prof.records
                      if name == "mi_policy_ws_bwd_gae_bf16"}
            assert all(T in s and N // MB in s for s in shapes), shapes
            assert int(ts.steps_taken) == (k + 1) * N * T
            # events: bit-exact under bf16 (they never pass through a GEMM)
            assert torch.equal(ts.env_states.obs.cpu(), ots.env_states.obs)
            done = info["rollout"].done
            assert 0.05 * N * T < int(done.sum()) < 0.5 * N * T    # resets do fire
            a, c, r = (info[n].numpy().mean() for n in ("actor", "critic", "regularization"))
            assert np.allclose(m["losses/actor/mean"].item(), a, rtol=5e-2, atol=5e-4), \
                (k, m["losses/actor/mean"].item(), a)
            assert np.allclose(m["losses/critic/mean"].item(), c, rtol=3e-3), \
                (k, m["losses/critic/mean"].item(), c)
            assert np.allclose(m["losses/regularization/mean"].item(), r, rtol=5e-2, atol=1e-4), \
                (k, m["losses/regularization/mean"].item(), r)
        norm, onorm = net.layers[0], onet.layers[0]
        assert float(norm.counter.value) == 2 * N * T == float(onorm.counter)
        assert np.allclose(norm.mean.value.cpu().numpy(), onorm.mean.numpy(), atol=1e-5)
        worst = 0.0
        for p, q in zip(net.parameters(), onet.parameters()):
            assert torch.isfinite(p.data).all()
            worst = max(worst, float((p.data.cpu() - q.detach()).abs().max()))
        assert worst < 2e-2, worst
        assert ops.handover_timeouts() == 0


def test_c2_benched_size_graph_equals_eager(dev):
    """Replays of the graph recorded at a size that contains the four benched launches, the
    in-kernel statistics hand-over and `mi_policy_loss_finalize_f32` == eager launches, bit
    for bit, and both stay within the oracle bars above (same seeds as that test)."""
    from nnx_ppo_amd import _lib, config, ops
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.loop import IterationRunner

    N, T, E, MB = 1024, 30, 2, 2
    outs = []
    with config.use_compute_dtype("bf16"):
        for graph in (True, False):
            env, _, net = _c2()
            ts = ppo.new_training_state(env, net, N, 23, 3e-4, device=dev)
            fn = lambda st, env=env: ppo.ppo_step(env, st, N, T, 0.95, 0.99, 0.2, True, False,
                                                  E, MB)
            r = IterationRunner(fn, ts, hip_graph=graph)
            with _lib.profiler as prof:          # iteration 1 is eager in both modes
                t = r.launch()
            assert BENCHED | {"mi_policy_loss_finalize_f32"} <= _called(prof)
            ms = [r.collect(t)]
            for _ in range(4):                   # 2 = recorded, 3-5 = replayed
                ms.append(r.collect(r.launch()))
            torch.cuda.synchronize()
            assert ("hip-graph" in r.launch_mode) == graph
            assert int(r.state.steps_taken) == 5 * N * T
            assert ops.handover_timeouts() == 0
            opt = r.state.optimizer
            outs.append((opt.params.clone(), opt.m.clone(), opt.v.clone(), ms,
                         net.layers[0].mean.value.clone()))
    (pa, ma, va, la, na), (pb, mb, vb, lb, nb) = outs
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert torch.equal(na, nb)
    for x, y in zip(la, lb):
        assert set(x) == set(y)
        for k in x:
            assert float(x[k]) == float(y[k]), k
    vals = [float(m["losses/critic/mean"]) for m in la]
    assert all(np.isfinite(v) for v in vals) and len(set(vals)) == 5


def test_c2_full_size_properties(dev):
    """BASELINE configs[1] at its full size through `train_ppo` on the replayed graph: steps
    counted, normaliser counter = k * T * N (`ppo_test.py:340-349`), every loss finite
    (`ppo_test.py:38-62`), no hand-over time-out, and the eager dispatch of this size is the
    bench's four launches."""
    from nnx_ppo_amd import _lib, config, ops
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.config import EvalConfig, PPOConfig, TrainConfig
    from nnx_ppo_amd.algorithms.types import LoggingLevel

    N, T, K = 4096, 30, 6
    env, _, net = _c2(max_steps=1000, max_len=1000)
    cfg = TrainConfig(
        ppo=PPOConfig(n_envs=N, rollout_length=T, total_steps=N * T * K, n_epochs=4,
                      n_minibatches=4, learning_rate=1e-4, logging_level=LoggingLevel.LOSSES),
        eval=EvalConfig(enabled=False), seed=17, checkpoint_every_steps=0)
    logs = []
    res = ppo.train_ppo(env, net, cfg, compute_dtype="bf16", hip_graph=True,
                        log_fn=lambda m, s: logs.append((s, {k: float(v) for k, v in m.items()})))
    assert res.total_iterations == K and res.total_steps == N * T * K
    assert int(res.training_state.steps_taken) == N * T * K
    assert int(res.training_state.optimizer.step) == 16 * K
    assert float(net.layers[0].counter.value) == K * N * T
    assert [s for s, _ in logs] == [N * T * i for i in range(1, K + 1)]
    for s, m in logs:
        for k, v in m.items():
            assert np.isfinite(v), (s, k, v)
        assert int(m["total_steps"]) == s
    crit = [m["losses/critic/mean"] for _, m in logs]
    assert len(set(crit)) == K                      # every replay did new work
    for p in net.parameters():
        assert torch.isfinite(p.data).all()
    assert ops.handover_timeouts() == 0
    ts = res.training_state
    with config.use_compute_dtype("bf16"), _lib.profiler as prof:
        ppo.ppo_step(env, ts, N, T, 0.95, 0.99, 0.2, True, False, 4, 4)
    used = _called(prof)
    assert BENCHED <= used, sorted(BENCHED - used)
    assert sum(1 for name, *_ in prof.records if name == "mi_policy_ws_bwd_gae_bf16") == 16
    assert sum(1 for name, *_ in prof.records if name == "mi_adam_step_slabs_f32") == 16


def test_handover_timeout_raises_in_its_iteration_and_writes_no_checkpoint(dev):
    """Force the bounded spin of `mi_policy_ws_bwd_gae_bf16`'s statistics hand-over to run out
    (test hook: 200 us, one arrival that never comes).  The sticky word rides in the
    iteration's metric copy, so `train_ppo` stops IN that iteration — before the checkpoint
    callback of that step sees the (NaN-poisoned) parameters.  Run once; the hook and the
    sticky words are cleared afterwards."""
    from nnx_ppo_amd import ops
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.config import EvalConfig, PPOConfig, TrainConfig
    from nnx_ppo_amd.algorithms.loop import InKernelTimeout
    from nnx_ppo_amd.algorithms.types import LoggingLevel

    N, T = 1024, 30
    env, _, net = _c2()
    cfg = TrainConfig(
        ppo=PPOConfig(n_envs=N, rollout_length=T, total_steps=N * T * 8, n_epochs=1,
                      n_minibatches=2, learning_rate=1e-4, logging_level=LoggingLevel.LOSSES),
        eval=EvalConfig(enabled=False), seed=5, checkpoint_every_steps=N * T)
    ckpts, logged = [], []

    def log_fn(m, s):
        logged.append(s)
        if s == 3 * N * T:      # checkpoints are due every step, so iteration 4 is not queued yet
            assert ops.set_handover_test_hook(limit_us=200, extra_arrivals=1) >= 1

    try:
        with pytest.raises(InKernelTimeout):
            ppo.train_ppo(env, net, cfg, compute_dtype="bf16", hip_graph=True, log_fn=log_fn,
                          checkpoint_fn=lambda ts, s: ckpts.append(
                              (s, bool(torch.isfinite(ts.optimizer.params).all()))))
        torch.cuda.synchronize()
        assert logged[-1] == 3 * N * T                      # iteration 4 never reached log_fn
        assert [s for s, _ in ckpts] == [0, N * T, 2 * N * T, 3 * N * T]
        assert all(ok for _, ok in ckpts)                   # every checkpoint written is clean
        assert ops.handover_timeouts() >= 1
        # the step that saw the time-out normalised with NaN: it cannot have trained silently
        assert not bool(torch.isfinite(torch.cat([p.data.flatten()
                                                  for p in net.parameters()])).all())
    finally:
        ops.set_handover_test_hook(0, 0)
        ops.clear_handover_timeouts()
        torch.cuda.synchronize()


def test_gae_loss_handover_timeout_is_counted_and_poisons(dev):
    """the stand-alone GAE + loss launch (`mi_gae_ppo_loss_f32`) had a bounded
    hand-over with no counter.  It now has the sticky word of the in-backward form, and a
    hand-over that ran out cannot produce finite gradients."""
    from nnx_ppo_amd import ops

    T, N = 30, 1024
    g = torch.Generator().manual_seed(4)
    r, v, lln, llo, reg = (torch.randn(T, N, generator=g).to(dev) for _ in range(5))
    lv = torch.randn(N, generator=g).to(dev)
    done = (torch.rand(T, N, generator=g) < 0.1).to(dev)
    trunc = (torch.rand(T, N, generator=g) < 0.05).to(dev)
    llo = lln + 0.05 * llo
    args = (r, v, lv, done, trunc, lln, llo, reg, 0.99, 0.95, True, 0.2, 1.0)
    g_ll0, g_v0, loss0, _ = ops.gae_ppo_loss(*args)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(g_ll0).all()) and ops.handover_timeouts() == 0
    assert any(w.numel() == 1 and w.dtype == torch.int32 for w in ops.health_words(dev))
    try:
        assert ops.set_handover_test_hook(limit_us=100, extra_arrivals=1) >= 1
        g_ll1, g_v1, _, _ = ops.gae_ppo_loss(*args)
        torch.cuda.synchronize()
        assert ops.handover_timeouts() >= 1
        assert sum(int(w.item()) for w in ops.health_words(dev)) >= 1
        assert not bool(torch.isfinite(g_ll1).all())
        assert torch.equal(g_v1, g_v0)        # the critic side does not use the statistics
    finally:
        ops.set_handover_test_hook(0, 0)
        ops.clear_handover_timeouts()
        torch.cuda.synchronize()
    g_ll2, _, _, _ = ops.gae_ppo_loss(*args)  # back to production behaviour, same bits
    assert torch.equal(g_ll2, g_ll0)
