"""The one-launch rollout (`mi_rollout_mock_ws_bf16`, csrc/rollout_ws.hip;
`MLPActorCritic.unroll_fused`) against the stepwise rollout it replaces (`unroll_env`,
`nnx_ppo/algorithms/rollout.py:48-73`: T x `single_transition`, rollout.py:11-45, with the
reset-on-done select of rollout.py:41-44 and EpisodeWrapper's counter / truncation,
`episode_wrapper.py:12-31`).  Every Transition leaf, the carried env state and the sampler's
noise bookkeeping must be BIT-IDENTICAL — the stepwise form is what the oracle tests pin
(tests/test_ppo_gpu.py, tests/test_benched_path_gpu.py), so the dispatch has to be invisible.
Cases: reset-heavy (inner done and wrapper truncation both fire), a batch that is not a
multiple of the 32-env tile, more tiles than workgroups, several trunk pairs, two rollouts in
a row (state hand-over), a whole `ppo_step`, and the oracle's event stream directly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(actor_h, critic_h, N, seed, max_steps, max_len, dev, obs=5, act=1):
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    env = EpisodeWrapper(MockEnv(obs, act, max_steps=max_steps), max_len)
    net = factories.make_mlp_actor_critic(obs, act, actor_h, critic_h, Rngs(seed))
    ts = ppo.new_training_state(env, net, N, seed, 1e-3, device=dev)
    return env, net, ts


def _same_tree(a, b, path="root"):
    """Structure and values: containers of the same type and keys, tensors bit-identical."""
    from nnx_ppo_amd.tree import TreeDataclass
    import dataclasses

    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        assert isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor), path
        assert a.shape == b.shape and a.dtype == b.dtype, (path, a.shape, b.shape, a.dtype, b.dtype)
        assert torch.equal(a, b), (path, (a != b).sum().item(), a.flatten()[:4], b.flatten()[:4])
        return 1
    if isinstance(a, dict):
        assert isinstance(b, dict) and set(a) == set(b), (path, list(a), list(b) if isinstance(b, dict) else b)
        return sum(_same_tree(a[k], b[k], f"{path}/{k}") for k in a)
    if isinstance(a, (list, tuple)):
        assert type(a) is type(b) and len(a) == len(b), (path, a, b)
        return sum(_same_tree(x, y, f"{path}[{i}]") for i, (x, y) in enumerate(zip(a, b)))
    if isinstance(a, TreeDataclass):
        assert type(a) is type(b), path
        return sum(_same_tree(getattr(a, f.name), getattr(b, f.name), f"{path}.{f.name}")
                   for f in dataclasses.fields(a))
    assert a == b or (a is None and b is None), (path, a, b)
    return 0


CASES = [
    # actor, critic, N, T, max_steps, max_len
    ([64] * 4, [256] * 2, 64, 9, 3, 5),        # C2's pair; the inner env ends every 3 steps
    ([64] * 4, [256] * 2, 64, 9, 7, 4),        # the wrapper truncates at 4, before the env ends
    ([64] * 4, [256] * 2, 100, 6, 4, 1000),    # ragged last tile
    ([64] * 4, [256] * 2, 4096, 30, 1000, 1000),   # BASELINE configs[1]: one tile per workgroup
    ([64] * 4, [256] * 2, 9000, 4, 2, 7),      # more tiles than workgroups: the tile loop
    ([64, 64], [64, 64], 96, 5, 4, 3),         # pair (64, 1, 64, 1)
    ([64, 64], [128, 128], 64, 8, 6, 50),      # pair (128, 1, 64, 1): tests/test_train_loop_gpu
    ([128, 128], [128, 128], 40, 5, 3, 4),
]


@pytest.mark.parametrize("actor_h,critic_h,N,T,max_steps,max_len", CASES)
def test_fused_rollout_equals_stepwise(dev, monkeypatch, actor_h, critic_h, N, T, max_steps,
                                       max_len):
    from nnx_ppo_amd import _lib, config
    from nnx_ppo_amd import random as rnd
    from nnx_ppo_amd.algorithms.rollout import unroll_env
    from nnx_ppo_amd.networks import policy

    out = []
    with config.use_compute_dtype("bf16"):
        for fused in (True, False):
            monkeypatch.setattr(policy, "FUSED_ROLLOUT", fused)
            env, net, ts = _make(actor_h, critic_h, N, 29, max_steps, max_len, dev)
            key = rnd.key(1234, device=dev)
            res = []
            net_state, env_state = ts.network_states, ts.env_states
            for it in range(2):      # the second rollout starts from the first one's state
                k = rnd.fold_in(key, it)
                with _lib.profiler as prof:
                    net_state, env_state, tr = unroll_env(env, env_state, net, net_state, T, k)
                used = {name for name, *_ in prof.records}
                assert ("mi_rollout_mock_ws_bf16" in used) == fused, used
                if fused:   # ONE launch (+ the first call's bf16 weight images and a host query)
                    assert used - {"mi_rollout_mock_ws_supported", "mi_weights_to_bf16_multi"} \
                        == {"mi_rollout_mock_ws_bf16"}, used
                    if it == 1:
                        assert used == {"mi_rollout_mock_ws_bf16"}, used
                res.append((net_state, env_state, tr))
            sampler = net.layers[-1].action.layers[-1]
            out.append((res, sampler._pending))
    (ra, pa), (rb, pb) = out
    assert pa == pb == 2 * T
    n_leaves = _same_tree(ra, rb)
    assert n_leaves >= 2 * 18   # 12 Transition leaves + 7 env-state leaves per rollout
    tr = ra[0][2]
    assert tr.obs.shape == (T, N, 5) and tr.done.dtype == torch.bool
    if max_steps < T:   # the reset path was exercised
        assert int(tr.done.sum()) > 0
    if max_len < max_steps and max_len <= T:
        assert int(tr.truncated.sum()) > 0


def test_fused_rollout_ppo_step_equals_stepwise(dev, monkeypatch):
    """Two whole iterations: parameters, moments, metrics and the state bit for bit."""
    from nnx_ppo_amd import config
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.networks import policy

    out = []
    with config.use_compute_dtype("bf16"):
        for fused in (True, False):
            monkeypatch.setattr(policy, "FUSED_ROLLOUT", fused)
            env, net, ts = _make([64] * 4, [256] * 2, 1024, 7, 6, 20, dev)
            ms = []
            for _ in range(2):
                ts, m = ppo.ppo_step(env, ts, 1024, 30, 0.95, 0.99, 0.2, True, False, 2, 2)
                ms.append({k: float(v) for k, v in m.items()})
            out.append((ts.optimizer.params.clone(), ts.optimizer.m.clone(), ms,
                        ts.env_states, ts.rng_key.clone(), int(ts.steps_taken)))
    (pa, ma, la, ea, ka, sa), (pb, mb, lb, eb, kb, sb) = out
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(ka, kb) and sa == sb
    assert la == lb
    _same_tree(ea, eb)


def test_fused_rollout_event_stream_vs_oracle(dev):
    """The one-launch rollout against the ORACLE's env, wrapper and key scheme directly
    (oracle/envs.py, oracle/keys.py): observations, flags and the carried state bit-exact."""
    from nnx_ppo_amd import _lib, config
    from nnx_ppo_amd.algorithms import ppo
    from oracle import envs as oe
    from oracle import keys as okeys
    from oracle import networks as on
    from oracle import ppo as op

    N, T = 256, 12
    with config.use_compute_dtype("bf16"):
        env, net, ts = _make([64] * 4, [256] * 2, N, 31, 5, 4, dev)
        oenv = oe.EpisodeWrapper(oe.MockEnv(5, 1, max_steps=5), 4)
        onet = on.from_product(net)
        ots = op.new_training_state(oenv, onet, N, 31, okeys, 1e-3)
        for k in range(2):
            with _lib.profiler as prof:
                ts, m = ppo.ppo_step(env, ts, N, T, 0.95, 0.99, 0.2, True, False, 1, 2)
            assert "mi_rollout_mock_ws_bf16" in {name for name, *_ in prof.records}
            ots, info = op.ppo_step(oenv, ots, N, T, 0.95, 0.99, 0.2, True, 1, 2, okeys)
            ro = info["rollout"]
            assert torch.equal(ts.env_states.obs.cpu(), ots.env_states.obs)
            assert torch.equal(ts.env_states.info["step_counter"].cpu(),
                               ots.env_states.info["step_counter"])
            assert torch.equal(ts.env_states.data["step_count"].cpu(),
                               ots.env_states.data["step_count"])
            assert int(ro.done.sum()) > N and int(ro.truncated.sum()) > 0
            c = info["critic"].numpy().mean()
            assert np.allclose(m["losses/critic/mean"].item(), c, rtol=3e-3)


def test_normal_observation_law_runs_stepwise_and_matches_the_oracle(dev):
    """`MockEnv(obs_law="normal")` (the reference mock's law, `mock_env.py:43,53`) is evaluated
    with torch ops: no fused env step, no one-launch rollout — and a whole `ppo_step` on it
    still matches the oracle: events exact, observations to the last ulp of log / cos."""
    from nnx_ppo_amd import _lib, config
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper
    from oracle import envs as oe
    from oracle import keys as okeys
    from oracle import networks as on
    from oracle import ppo as op

    N, T = 128, 8
    with config.use_compute_dtype("bf16"):
        env = EpisodeWrapper(MockEnv(5, 1, max_steps=5, obs_law="normal"), 20)
        oenv = oe.EpisodeWrapper(oe.MockEnv(5, 1, max_steps=5, obs_law="normal"), 20)
        net = factories.make_mlp_actor_critic(5, 1, [64] * 4, [256] * 2, Rngs(3))
        ts = ppo.new_training_state(env, net, N, 3, 1e-3, device=dev)
        onet = on.from_product(net)
        ots = op.new_training_state(oenv, onet, N, 3, okeys, 1e-3)
        with _lib.profiler as prof:
            ts, m = ppo.ppo_step(env, ts, N, T, 0.95, 0.99, 0.2, True, False, 1, 2)
        used = {name for name, *_ in prof.records}
        assert "mi_rollout_mock_ws_bf16" not in used and "mi_mock_env_step" not in used
        ots, info = op.ppo_step(oenv, ots, N, T, 0.95, 0.99, 0.2, True, 1, 2, okeys)
        assert np.allclose(ts.env_states.obs.cpu().numpy(), ots.env_states.obs.numpy(),
                           rtol=0, atol=4e-6)
        assert torch.equal(ts.env_states.data["step_count"].cpu(),
                           ots.env_states.data["step_count"])
        assert torch.equal(ts.env_states.info["step_counter"].cpu(),
                           ots.env_states.info["step_counter"])
        c = info["critic"].numpy().mean()
        assert np.allclose(m["losses/critic/mean"].item(), c, rtol=3e-3)


@pytest.mark.parametrize("H,critic_h,N,T,max_steps,max_len",
                         [(64, [256, 256], 64, 10, 3, 50), (64, [256, 256], 100, 7, 6, 4),
                          (64, [256, 256], 4096, 30, 5, 1000), (128, [128, 128], 96, 6, 2, 9),
                          (64, [64, 64], 9000, 3, 2, 7)])
def test_fused_gru_rollout_equals_stepwise(dev, monkeypatch, H, critic_h, N, T, max_steps,
                                           max_len):
    """`mi_rollout_mock_gru_ws_bf16` (the recurrent actor of make_gru_actor_critic with its
    carry on chip for the T steps, reset to zeros on done) against the stepwise rollout
    (mi_gru_policy_step_bf16 + mi_mock_episode_step_select + the carry's reset select):
    Transition, env state AND the carry bit-identical, two rollouts in a row."""
    from nnx_ppo_amd import _lib, config
    from nnx_ppo_amd import random as rnd
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.rollout import unroll_env
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks import factories, policy
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    out = []
    with config.use_compute_dtype("bf16"):
        for fused in (True, False):
            monkeypatch.setattr(policy, "FUSED_ROLLOUT", fused)
            env = EpisodeWrapper(MockEnv(5, 1, max_steps=max_steps), max_len)
            net = factories.make_gru_actor_critic(5, 1, H, critic_h, Rngs(13))
            ts = ppo.new_training_state(env, net, N, 13, 1e-3, device=dev)
            key = rnd.key(77, device=dev)
            res = []
            net_state, env_state = ts.network_states, ts.env_states
            for it in range(2):
                with _lib.profiler as prof:
                    net_state, env_state, tr = unroll_env(env, env_state, net, net_state, T,
                                                          rnd.fold_in(key, it))
                used = {name for name, *_ in prof.records}
                assert ("mi_rollout_mock_gru_ws_bf16" in used) == fused, used
                if fused and it == 1:
                    assert used == {"mi_rollout_mock_gru_ws_bf16"}, used
                res.append((net_state, env_state, tr))
            out.append(res)
    ra, rb = out
    assert _same_tree(ra, rb) >= 2 * 19     # + the carry
    h = ra[1][0][-1]["action"][1]
    assert h.shape == (N, H)
    done_last = ra[1][2].done[-1]
    assert float(h[done_last].abs().sum()) == 0.0   # reset rows carry zeros
    if bool((~done_last).any()):
        assert float(h[~done_last].abs().sum()) > 0  # the others carry their state
    assert int(ra[0][2].done.sum()) > 0


def test_fused_gru_rollout_ppo_step_equals_stepwise(dev, monkeypatch):
    from nnx_ppo_amd import config
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import cartpole_shaped
    from nnx_ppo_amd.networks import factories, policy
    from nnx_ppo_amd.networks.types import Rngs

    out = []
    with config.use_compute_dtype("bf16"):
        for fused in (True, False):
            monkeypatch.setattr(policy, "FUSED_ROLLOUT", fused)
            env = cartpole_shaped(max_steps=5)
            from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper
            env = EpisodeWrapper(env, 1000)
            net = factories.make_gru_actor_critic(5, 1, 64, [256, 256], Rngs(42))
            ts = ppo.new_training_state(env, net, 512, 42, 3e-4, device=dev)
            ms = []
            for _ in range(2):
                ts, m = ppo.ppo_step(env, ts, 512, 30, 0.95, 0.99, 0.2, True, False, 2, 2)
                ms.append({k: float(v) for k, v in m.items()})
            out.append((ts.optimizer.params.clone(), ms, ts.network_states, ts.env_states))
    (pa, la, na, ea), (pb, lb, nb, eb) = out
    assert torch.equal(pa, pb) and la == lb
    _same_tree(na, nb)
    _same_tree(ea, eb)
