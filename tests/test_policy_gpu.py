"""One-launch policy evaluation (`mi_policy_fwd_bf16` / `mi_policy_bwd_bf16`,
networks/policy.py) against the
generic container path it replaces: the same normaliser expression, the same trunk
code and the same sampler row function, so every output — actions, log-likelihoods,
values, regulariser, extras, saved images, gradients — must be BIT-identical."""
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


def _net(dev, obs, act_dim, actor_h, critic_h, activation, seed=3):
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    net = factories.make_mlp_actor_critic(obs, act_dim, actor_h, critic_h, Rngs(seed),
                                          activation=activation)
    net.to(dev)
    # non-trivial normaliser statistics
    n = net.layers[0]
    g = torch.Generator().manual_seed(seed)
    n.mean.value.copy_(torch.randn(obs, generator=g))
    n.M2.value.copy_(torch.rand(obs, generator=g) * 50 + 1)
    n.counter.value.fill_(37.0)
    return net


def _sampler(net):
    return net.layers[-1].action.layers[-1]


def _leaves(x):
    from nnx_ppo_amd.tree import tree_leaves

    return [t for t in tree_leaves(x) if isinstance(t, torch.Tensor)]


@pytest.mark.parametrize("M", [1, 100, 4096, 9000])
@pytest.mark.parametrize("activation", ["relu", "tanh", "swish"])
# third shape: a NARROW action trunk (<= 64 wide: 256 rows per workgroup above 8192 rows)
# whose widths are not multiples of 32 (pad tiles, general epilogue) with a 6-column head
@pytest.mark.parametrize("shape", [(5, 1, [64] * 4, [256] * 2), (17, 6, [256] * 2, [128]),
                                   (7, 3, [48, 40], [96])])
def test_rollout_call_is_bit_identical_to_generic(dev, bf16, M, activation, shape):
    from nnx_ppo_amd.networks.containers import Sequential
    from nnx_ppo_amd.networks.policy import MLPActorCritic

    obs_dim, act_dim, ah, ch = shape
    net = _net(dev, obs_dim, act_dim, ah, ch, activation)
    assert isinstance(net, MLPActorCritic)
    x = torch.randn(M, obs_dim, generator=torch.Generator().manual_seed(M)).to(dev)
    state = net.initialize_state(M)
    smp = _sampler(net)
    for deterministic in (False, True):
        smp.deterministic = deterministic
        smp._pending = 5
        fused = net(state, x)
        smp._pending = 5
        plain = Sequential.__call__(net, state, x)
        for name in ("next_state", "output", "regularization_loss", "metrics", "rollout_extras"):
            a, b = _leaves(getattr(fused, name)), _leaves(getattr(plain, name))
            assert len(a) == len(b) and len(a) > 0 or name == "next_state", name
            for u, v in zip(a, b):
                assert u.shape == v.shape and torch.equal(u, v), name
    torch.cuda.synchronize()


@pytest.mark.parametrize("T,B,shape", [(3, 7, (5, 2, [64, 64], [256, 256])),
                                       (30, 1024, (5, 2, [64, 64], [256, 256])),
                                       # narrow trunk, ragged last workgroup, odd widths
                                       (9, 1000, (7, 3, [48, 40], [96]))])
@pytest.mark.parametrize("activation", ["relu", "swish"])
def test_replay_and_gradients_bit_identical_to_generic(dev, bf16, T, B, shape, activation):
    from nnx_ppo_amd.networks.containers import Sequential
    from nnx_ppo_amd.networks.types import PPONetworkOutput, bump_param_epoch
    from nnx_ppo_amd.optim import Optimizer

    obs_dim = shape[0]
    net = _net(dev, *shape, activation)
    opt = Optimizer(net, 1e-3, None, None, device=dev)
    g = torch.Generator().manual_seed(T * B)
    x_seq = torch.randn(T, B, obs_dim, generator=g).to(dev)
    done = torch.zeros(T, B, dtype=torch.bool, device=dev)
    state = net.initialize_state(B)
    smp = _sampler(net)
    # rollout extras as a rollout would have stacked them
    steps = [net(state, x_seq[t]).rollout_extras for t in range(T)]
    from nnx_ppo_amd.tree import tree_map

    extras = tree_map(lambda *xs: torch.stack(xs, 0), steps[0], *steps[1:])
    g_ll = torch.randn(T, B, generator=g).to(dev)
    g_v = torch.randn(T, B, generator=g).to(dev)
    res = []
    for fused in (True, False):
        bump_param_epoch()
        opt.begin()
        smp._pending = 11
        if fused:
            ctx, out, reg, fs = net.replay(state, x_seq, done, extras, need_input_grad=False)
            assert ctx[0] == "fused"
            net.replay_backward(ctx, PPONetworkOutput(None, g_ll, g_v), 1.0 / (T * B))
        else:
            ctx, out, reg, fs = Sequential.replay(net, state, x_seq, done, extras,
                                                  need_input_grad=False)
            Sequential.replay_backward(net, ctx, PPONetworkOutput(None, g_ll, g_v),
                                       1.0 / (T * B))
        torch.cuda.synchronize()
        res.append((out.loglikelihoods.clone(), out.value_estimates.clone(), reg.clone(),
                    opt.grads.clone()))
    for a, b in zip(res[0][:3], res[1][:3]):
        assert a.shape == b.shape and torch.equal(a, b)
    # gradients: the same bf16 operands (dz, x) in both paths, but the fused path sums
    # the row splits of ALL layers' dW in one grouped launch, i.e. in a different fp32
    # order: equal to accumulation-order rounding
    ga, gb = res[0][3], res[1][3]
    assert float(ga.abs().sum()) > 0
    scale = float(gb.abs().max())
    assert float((ga - gb).abs().max()) <= 2e-5 * scale


def test_bootstrap_rows_ride_along_bit_identically(dev, bf16):
    """`replay_with_bootstrap`: the value of the bootstrap observation from the value
    trunk's tail rows == `forward_value`, and everything else == `replay`, bit for bit."""
    from nnx_ppo_amd.networks.types import PPONetworkOutput, bump_param_epoch
    from nnx_ppo_amd.optim import Optimizer
    from nnx_ppo_amd.tree import tree_map

    T, B = 30, 1024
    net = _net(dev, 5, 1, [64] * 4, [256] * 2, "relu")
    opt = Optimizer(net, 1e-3, None, None, device=dev)
    g = torch.Generator().manual_seed(1)
    x_seq = torch.randn(T, B, 5, generator=g).to(dev)
    last_obs = torch.randn(B, 5, generator=g).to(dev)
    done = torch.zeros(T, B, dtype=torch.bool, device=dev)
    state = net.initialize_state(B)
    smp = _sampler(net)
    steps = [net(state, x_seq[t]).rollout_extras for t in range(T)]
    extras = tree_map(lambda *xs: torch.stack(xs, 0), steps[0], *steps[1:])
    g_ll = torch.randn(T, B, generator=g).to(dev)
    g_v = torch.randn(T, B, generator=g).to(dev)
    res = []
    for with_boot in (True, False):
        bump_param_epoch()
        opt.begin()
        smp._pending = 3
        if with_boot:
            ctx, out, reg, fs, lv = net.replay_with_bootstrap(state, x_seq, done, extras, last_obs)
        else:
            ctx, out, reg, fs = net.replay(state, x_seq, done, extras, need_input_grad=False)
            lv = net.forward_value(state, last_obs)
        assert ctx[0] == "fused" and lv.shape == (B,)
        net.replay_backward(ctx, PPONetworkOutput(None, g_ll, g_v), 1.0 / (T * B))
        torch.cuda.synchronize()
        res.append((out.loglikelihoods.clone(), out.value_estimates.clone(), reg.clone(),
                    lv.clone(), opt.grads.clone()))
    for a, b in zip(*res):
        assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("M", [1, 100, 8192])
@pytest.mark.parametrize("shape", [({"position": 8, "velocity": 9}, 6, [256] * 4, [512] * 2),
                                   ({"b": 3, "a": 2, "c": 1}, 2, [64] * 2, [64] * 2)])
def test_pytree_obs_rollout_call_is_bit_identical_to_generic(dev, bf16, M, shape):
    """`make_mlp_actor_critic` with a dict `obs_size` builds Sequential([Normalizer(tree),
    Flattener, PPOAdapter]) (BASELINE config 3's shape); its rollout step is one
    concatenation + one launch, bit-identical to the generic containers — per-leaf
    statistics, sorted-key leaf order, the raw PyTree kept as the normaliser's extras."""
    from nnx_ppo_amd import _lib
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.containers import Sequential
    from nnx_ppo_amd.networks.policy import MLPActorCritic
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.tree import tree_leaves

    obs_tree, act_dim, ah, ch = shape
    net = factories.make_mlp_actor_critic(obs_tree, act_dim, ah, ch, Rngs(4))
    assert isinstance(net, MLPActorCritic) and len(net.layers) == 3
    net.to(dev)
    g = torch.Generator().manual_seed(1)
    n = net.layers[0]
    for k, w in obs_tree.items():
        n.mean.value[k].copy_(torch.randn(w, generator=g))
        n.M2.value[k].copy_(torch.rand(w, generator=g) * 50 + 1)
    n.counter.value.fill_(37.0)
    # insertion order differs from sorted order on purpose
    x = {k: torch.randn(M, w, generator=g).to(dev) for k, w in reversed(list(obs_tree.items()))}
    state = net.initialize_state(M)
    smp = _sampler(net)
    for deterministic in (False, True):
        smp.deterministic = deterministic
        smp._pending = 5
        with _lib.profiler as prof:
            fused = net(state, x)
        used = [r[0] for r in prof.records]
        assert sum(u in ("mi_policy_fwd_bf16", "mi_policy_ws_fwd_bf16") for u in used) == 1
        assert "mi_normalize_fwd_f32" not in used and "mi_mlp_fwd_bf16" not in used
        smp._pending = 5
        plain = Sequential.__call__(net, state, x)
        for name in ("next_state", "output", "regularization_loss", "metrics", "rollout_extras"):
            a, b = _leaves(getattr(fused, name)), _leaves(getattr(plain, name))
            assert len(a) == len(b), name
            for u, v in zip(a, b):
                assert u.shape == v.shape and torch.equal(u, v), name
        assert set(fused.rollout_extras[0]) == set(obs_tree)
    # the statistics stay per-leaf tensors the normaliser updates in place: one update, then
    # the fused step still equals the generic one
    seq = {k: torch.randn(3, M, w, generator=g).to(dev) for k, w in obs_tree.items()}
    net.layers[0].update_statistics(seq)
    assert all(t.dim() == 1 for t in tree_leaves(n.mean.value))
    smp._pending = 9
    fused = net(state, x)
    smp._pending = 9
    plain = Sequential.__call__(net, state, x)
    for u, v in zip(_leaves(fused.output), _leaves(plain.output)):
        assert torch.equal(u, v)
    # a replaced leaf (checkpoint load, .to()) is picked up
    n.mean.value = {k: v.clone() + 1.0 for k, v in n.mean.value.items()}
    smp._pending = 11
    fused = net(state, x)
    smp._pending = 11
    plain = Sequential.__call__(net, state, x)
    for u, v in zip(_leaves(fused.output), _leaves(plain.output)):
        assert torch.equal(u, v)


@pytest.mark.parametrize("T,B", [(3, 7), (30, 1024)])
def test_pytree_obs_replay_with_bootstrap_equals_generic(dev, bf16, T, B):
    """The loss replay of a PyTree-observation network in the fused shape class: sequence
    tree and bootstrap observation are concatenated once each and take the one-launch
    replay (+ bootstrap rows); outputs bit-identical to the generic containers, gradients
    to dW summation order."""
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.containers import Sequential
    from nnx_ppo_amd.networks.types import PPONetworkOutput, Rngs, bump_param_epoch
    from nnx_ppo_amd.optim import Optimizer
    from nnx_ppo_amd.tree import tree_map

    obs_tree = {"position": 8, "velocity": 9}
    net = factories.make_mlp_actor_critic(obs_tree, 3, [128] * 2, [256] * 2, Rngs(5))
    net.to(dev)
    opt = Optimizer(net, 1e-3, None, None, device=dev)
    g = torch.Generator().manual_seed(T * B)
    n = net.layers[0]
    for k, w in obs_tree.items():
        n.mean.value[k].copy_(torch.randn(w, generator=g))
        n.M2.value[k].copy_(torch.rand(w, generator=g) * 50 + 1)
    n.counter.value.fill_(11.0)
    x_seq = {k: torch.randn(T, B, w, generator=g).to(dev) for k, w in obs_tree.items()}
    last = {k: torch.randn(B, w, generator=g).to(dev) for k, w in obs_tree.items()}
    done = torch.zeros(T, B, dtype=torch.bool, device=dev)
    state = net.initialize_state(B)
    smp = _sampler(net)
    steps = [net(state, {k: v[t] for k, v in x_seq.items()}).rollout_extras for t in range(T)]
    extras = tree_map(lambda *xs: torch.stack(xs, 0), steps[0], *steps[1:])
    g_ll = torch.randn(T, B, generator=g).to(dev)
    g_v = torch.randn(T, B, generator=g).to(dev)
    res = []
    for fused in (True, False):
        bump_param_epoch()
        opt.begin()
        smp._pending = 11
        if fused:
            ctx, out, reg, fs, lv = net.replay_with_bootstrap(state, x_seq, done, extras, last)
            assert ctx[0] == "fused" and len(fs) == 3
            net.replay_backward(ctx, PPONetworkOutput(None, g_ll, g_v), 1.0 / (T * B))
        else:
            ctx, out, reg, fs = Sequential.replay(net, state, x_seq, done, extras,
                                                  need_input_grad=False)
            lv = Sequential.forward_value(net, state, last)
            Sequential.replay_backward(net, ctx, PPONetworkOutput(None, g_ll, g_v),
                                       1.0 / (T * B))
        torch.cuda.synchronize()
        res.append((out.loglikelihoods.clone(), out.value_estimates.clone(), reg.clone(),
                    lv.reshape(-1).clone(), opt.grads.clone()))
    for a, b in zip(res[0][:4], res[1][:4]):
        assert a.shape == b.shape and torch.equal(a, b)
    ga, gb = res[0][4], res[1][4]
    assert float(ga.abs().sum()) > 0
    assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max())
