"""The recurrent actor-critic's rollout step as ONE launch (`mi_gru_policy_step_bf16`,
networks/policy.py:GRUActorCritic) against the generic containers it replaces — normaliser,
Dense, the GRU's projection + recurrent step (gru_mfma.hip), Dense, sampler, value trunk:
the same operand roundings, k-order and expressions, so every output must be BIT-identical."""
import pytest
import torch

from nnx_ppo_amd import config

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bf16():
    prev = config.compute_dtype()
    config.set_compute_dtype("bf16")
    yield
    config.set_compute_dtype(prev)


def _net(dev, obs, act, H, critic_h, seed=3):
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    net = factories.make_gru_actor_critic(obs, act, H, critic_h, Rngs(seed))
    net.to(dev)
    n = net.layers[0]
    g = torch.Generator().manual_seed(seed)
    n.mean.value.copy_(torch.randn(obs, generator=g))
    n.M2.value.copy_(torch.rand(obs, generator=g) * 50 + 1)
    n.counter.value.fill_(37.0)
    # non-zero biases everywhere (they initialise to zero)
    for p in net.parameters():
        if len(p.shape) == 1:
            p.data.copy_(torch.randn(p.shape[0], generator=g) * 0.3)
    return net


def _leaves(x):
    from nnx_ppo_amd.tree import tree_leaves

    return [t for t in tree_leaves(x) if isinstance(t, torch.Tensor)]


@pytest.mark.parametrize("M", [1, 100, 4096])
@pytest.mark.parametrize("shape", [(5, 1, 64, [256, 256]), (17, 3, 128, [128, 128]),
                                   (9, 2, 64, [64, 64])])
def test_gru_rollout_step_is_bit_identical_to_generic(dev, bf16, M, shape):
    from nnx_ppo_amd import _lib
    from nnx_ppo_amd.networks.containers import Sequential
    from nnx_ppo_amd.networks.policy import GRUActorCritic

    obs_dim, act_dim, H, ch = shape
    net = _net(dev, obs_dim, act_dim, H, ch)
    assert isinstance(net, GRUActorCritic)
    g = torch.Generator().manual_seed(M)
    smp = net.layers[-1].action.layers[-1]
    state_f = state_p = net.initialize_state(M)
    # a non-zero carry to start from
    h0 = torch.randn(M, H, generator=g).to(dev)
    state_f = [(), {"action": [(), h0.clone(), (), ()], "value": list(state_f[-1]["value"])}]
    state_p = [(), {"action": [(), h0.clone(), (), ()], "value": list(state_p[-1]["value"])}]
    for step in range(3):  # the carry is threaded through
        x = torch.randn(M, obs_dim, generator=g).to(dev)
        smp.deterministic = step == 2
        smp._pending = 5 + step
        with _lib.profiler as prof:
            fused = net(state_f, x)
        assert "mi_gru_policy_step_bf16" in [r[0] for r in prof.records]
        assert not any(r[0].startswith("mi_gru_seq") for r in prof.records)
        smp._pending = 5 + step
        plain = Sequential.__call__(net, state_p, x)
        for name in ("next_state", "output", "regularization_loss", "metrics", "rollout_extras"):
            a, b = _leaves(getattr(fused, name)), _leaves(getattr(plain, name))
            assert len(a) == len(b) and (len(a) > 0), name
            for u, v in zip(a, b):
                assert u.shape == v.shape and torch.equal(u, v), (name, step)
        state_f, state_p = fused.next_state, plain.next_state
    torch.cuda.synchronize()


def test_gru_ppo_step_fused_rollout_equals_generic(dev, bf16):
    """A whole iteration (rollout with resets, replay, BPTT, update) with the one-launch
    rollout step == the same iteration through the generic containers."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks import factories, policy
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    outs = []
    for fused in (True, False):
        policy.FUSED = fused
        try:
            env = EpisodeWrapper(MockEnv(5, 1, max_steps=5), 40)
            net = factories.make_gru_actor_critic(5, 1, 64, [256, 256], Rngs(17))
            ts = ppo.new_training_state(env, net, 256, 17, 1e-3, device=dev)
            for _ in range(2):
                ts, m = ppo.ppo_step(env, ts, 256, 10, 0.95, 0.99, 0.2, True, False, 2, 2)
            outs.append((ts.optimizer.params.clone(), {k: float(v) for k, v in m.items()},
                         [t.clone() for t in _leaves(ts.network_states)]))
        finally:
            policy.FUSED = True
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]
    for a, b in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, b)


def test_gru_step_support_query(dev):
    from nnx_ppo_amd import ops

    R, N = ops.ACT_RELU, ops.ACT_NONE
    assert ops.gru_policy_step_supported(5, 64, 2, [5, 256, 256, 1], [R, R, N])
    assert ops.gru_policy_step_supported(17, 128, 6, [17, 128, 128, 1], [R, R, N])
    assert not ops.gru_policy_step_supported(5, 96, 2, [5, 256, 256, 1], [R, R, N])   # width
    assert not ops.gru_policy_step_supported(40, 64, 2, [40, 256, 256, 1], [R, R, N])  # K0 <= 32
    assert not ops.gru_policy_step_supported(5, 64, 2, [5, 512, 1], [R, N])           # value trunk


@pytest.mark.parametrize("n_envs,T", [(256, 10), (2048, 30)])
def test_bootstrap_rows_in_the_value_chain_equal_forward_value(dev, bf16, n_envs, T):
    """`Sequential.replay_with_bootstrap` (the bootstrap observation as step T of the
    stateless value chain's replay, ppo.py:433-437) == `replay` + `forward_value`: a row of a
    Dense chain does not depend on its neighbours, so two whole iterations — parameters,
    metrics, carries — must be BIT-identical.  (2048 x 30: the value chain on the
    weights-stationary kernels, its backward over a row prefix of the forward's images.)"""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks import containers, factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    outs = []
    for on in (True, False):
        containers.BOOTSTRAP_IN_CHAIN = on
        try:
            env = EpisodeWrapper(MockEnv(5, 1, max_steps=5), 40)
            net = factories.make_gru_actor_critic(5, 1, 64, [256, 256], Rngs(17))
            ts = ppo.new_training_state(env, net, n_envs, 17, 1e-3, device=dev)
            for _ in range(2):
                ts, m = ppo.ppo_step(env, ts, n_envs, T, 0.95, 0.99, 0.2, True, False, 2, 2)
            outs.append((ts.optimizer.params.clone(), {k: float(v) for k, v in m.items()},
                         [t.clone() for t in _leaves(ts.network_states)]))
        finally:
            containers.BOOTSTRAP_IN_CHAIN = True
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]
    for a, b in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, b)


def test_gru_bias_tail_gradient_folded_equals_reduced(dev, bf16):
    """The GRU recurrent kernel's dW request names `b_hn.grad` with `bias_first = 2H`.  With
    the step's slabs folded into the optimiser launch the n gate's column sums are picked out
    of the slabs (`mi_adam_step_slabs_f32`, gb_first); when something reads the gradient first
    (GRAD_NORM logging: `flush_pending_slabs`) the 3H sums go to a scratch vector whose tail
    is added.  Same summation order: two iterations must leave BIT-identical parameters."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.config import LoggingLevel
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    outs = []
    for level in (LoggingLevel.LOSSES, LoggingLevel.LOSSES | LoggingLevel.GRAD_NORM):
        env = EpisodeWrapper(MockEnv(5, 1, max_steps=5), 40)
        net = factories.make_gru_actor_critic(5, 1, 64, [256, 256], Rngs(17))
        ts = ppo.new_training_state(env, net, 256, 17, 1e-3, device=dev)
        for _ in range(2):
            ts, m = ppo.ppo_step(env, ts, 256, 10, 0.95, 0.99, 0.2, True, False, 2, 2,
                                 logging_level=level)
        outs.append(ts.optimizer.params.clone())
        assert all(torch.isfinite(torch.as_tensor(v)).all() for v in m.values())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("H,n_envs,T,A", [(64, 512, 30, 1), (32, 148, 12, 3), (96, 264, 9, 2),
                                          (128, 512, 10, 1), (64, 128, 40, 1)])
def test_gru_replay_with_head_and_sampler_in_the_sequence_launch(dev, monkeypatch, H, n_envs, T, A):
    """`mi_gru_seq_fwd_tail_bf16`: the linear head and the sampler's replay behind the GRU ride
    in the sequence launch (containers.Sequential.replay, REC_TAIL).  Against the launches they
    replace — same network, same rollout, same minibatch: every loss scalar and every parameter
    gradient bit for bit (the head's MFMA tiles, k order and the sampler's row function are the
    same), and the launch list.  The same for the backward mirror, `mi_gru_seq_bwd_tail_bf16`
    (sampler backward + the head's dX in front of the BPTT, REC_TAIL_BWD), and for the input
    projection inside both (`mi_gru_seq_fwd_proj_tail_bf16` / `mi_gru_seq_bwd_proj_tail_bf16`,
    REC_PROJ: gi = y W_i + b_i per step from the bf16 image of the relu layer in front, its
    backward behind the BPTT step), and for the relu Dense in front of the projection
    (`mi_gru_seq_fwd_front_proj_tail_bf16`, REC_FRONT).  Every state width of the matrix-core GRU (H = 32 and 96:
    waves without a unit tile; 128: two tiles per wave), ragged batches (74 and 132 rows per
    minibatch) and several action sizes."""
    from nnx_ppo_amd import _lib, config
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks import containers, factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    from nnx_ppo_amd import ops

    # (T = 40: the front layer's rows no longer fit in LDS beside the history of h — the
    # projection still rides, the layer keeps its own launch)
    front_fits = ops.gru_seq_front_supported(T, n_envs // 2, H, 5, 2 * A)
    assert front_fits == (T != 40)
    out, n_launch = [], []
    with config.use_compute_dtype("bf16"):
        for tail, tail_bwd, proj, front in ((True, True, True, True), (True, True, True, False),
                                            (True, True, False, False),
                                            (True, False, False, False),
                                            (False, False, False, False)):
            monkeypatch.setattr(containers, "REC_TAIL", tail)
            monkeypatch.setattr(containers, "REC_TAIL_BWD", tail_bwd)
            monkeypatch.setattr(containers, "REC_PROJ", proj)
            monkeypatch.setattr(containers, "REC_FRONT", front)
            env = EpisodeWrapper(MockEnv(5, A, max_steps=5), 1000)
            net = factories.make_gru_actor_critic(5, A, H, [256, 256], Rngs(9))
            ts = ppo.new_training_state(env, net, n_envs, 9, 3e-4, device=dev)
            ms = []
            for it in range(2):
                with _lib.profiler as prof:
                    ts, m = ppo.ppo_step(env, ts, n_envs, T, 0.95, 0.99, 0.2, True, False, 2, 2)
                ms.append({k: float(v) for k, v in m.items()})
                used = {name for name, *_ in prof.records}
                assert ("mi_gru_seq_fwd_tail_bf16" in used) == (tail and not proj), used
                # ... and `mi_gru_seq_*_proj_tail_bf16` the input projection too (REC_PROJ)
                assert ("mi_gru_seq_fwd_proj_tail_bf16" in used) == \
                    (proj and not (front and front_fits)), used
                # ... and the relu Dense in front of it (REC_FRONT)
                assert ("mi_gru_seq_fwd_front_proj_tail_bf16" in used) == (front and front_fits), used
                assert ("mi_gru_seq_bwd_proj_tail_bf16" in used) == proj, used
                if tail and H == 64:  # (other widths roll out through the generic containers)
                    assert "mi_tanh_gauss_fwd_f32" not in used and "mi_gru_seq_fwd_bf16" not in used
                # ... and `mi_gru_seq_bwd_tail_bf16` their backward (REC_TAIL_BWD)
                assert ("mi_gru_seq_bwd_tail_bf16" in used) == (tail_bwd and not proj), used
                if tail_bwd:
                    assert not {"mi_tanh_gauss_bwd_f32", "mi_gru_seq_bwd_bf16"} & used, used
            n_launch.append(sum(not name.endswith("_supported") for name, *_ in prof.records))
            out.append((ts.optimizer.params.clone(), ts.optimizer.m.clone(), ms))
    # 2 epochs x 2 minibatches: the front layer's forward launch goes, the chain's backward launch
    # goes with the projection inside, then two launches fewer per gradient step each time
    assert [b - a for a, b in zip(n_launch, n_launch[1:])] == \
        [4 if front_fits else 0, 4, 8, 8], n_launch
    for (pa, ma, la), (pb, mb, lb) in zip(out, out[1:]):
        assert la == lb
        assert torch.equal(pa, pb) and torch.equal(ma, mb)
