"""LSTM carry — the reference's recurrent layer (`nnx_ppo/networks/recurrent.py:16-161`).
Contract tests restate `recurrent_test.py`: zeros init / zeros-like reset (59-93),
minibatch slicing == full batch (184-211), regularisation shape (213-230), an
LSTM actor through rollout / `ppo_step` with resets every 5 steps and gradient clipping
1.0 (232-343).  The cell arithmetic is flax's LSTMCell (third-party, PARITY UNPINNED by
the reference): pinned here against the oracle's restatement and fp64 autograd."""
import numpy as np
import pytest
import torch

from nnx_ppo_amd import random as keys
from oracle import networks as on
from oracle import ppo as op

pytestmark = pytest.mark.gpu
D = torch.float64


def _lstm(in_f, H, seed=0):
    from nnx_ppo_amd.networks.recurrent import LSTM
    from nnx_ppo_amd.networks.types import Rngs

    m = LSTM(in_f, H, Rngs(seed))
    rng = np.random.default_rng(seed + 1)
    m.b_h.data = torch.tensor(rng.normal(0, 0.1, size=4 * H), dtype=torch.float32)
    return m


@pytest.mark.parametrize("T,B,I,H", [(1, 1, 3, 16), (30, 64, 64, 64), (12, 37, 5, 64),
                                     (7, 100, 16, 32), (5, 33, 8, 128), (6, 20, 7, 80),
                                     (4, 2051, 5, 64), (3, 4099, 5, 64)])
def test_lstm_sequence_fwd_bwd_vs_oracle(dev, T, B, I, H):
    from nnx_ppo_amd.optim import Optimizer

    m = _lstm(I, H, seed=T + B)
    m.to(dev)
    opt = Optimizer(m, 1e-3, device=dev)
    om = on.from_product(m)
    rng = np.random.default_rng(B)
    x = rng.normal(size=(T, B, I)).astype(np.float32)
    h0 = rng.normal(size=(B, H)).astype(np.float32)
    c0 = rng.normal(size=(B, H)).astype(np.float32)
    done = rng.random((T, B)) < 0.2
    gy = rng.normal(size=(T, B, H)).astype(np.float32)
    t = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt).to(dev)
    ctx, out, reg, (h_f, c_f) = m.replay((t(h0), t(c0)), t(x), t(done, torch.bool), None,
                                          need_input_grad=True)
    assert reg is None and out.shape == (T, B, H)
    # oracle: step-wise scan with reset-on-done (ppo.py:411-418)
    x64 = torch.tensor(x, dtype=D, requires_grad=True)
    h, c = torch.tensor(h0, dtype=D), torch.tensor(c0, dtype=D)
    outs = []
    for k in range(T):
        o = om((h, c), x64[k])
        outs.append(o.output)
        d = torch.tensor(done[k])[:, None]
        h = torch.where(d, torch.zeros_like(o.next_state[0]), o.next_state[0])
        c = torch.where(d, torch.zeros_like(o.next_state[1]), o.next_state[1])
    want = torch.stack(outs)
    assert np.allclose(out.cpu().numpy(), want.detach().numpy(), rtol=1e-4, atol=2e-5)
    assert np.allclose(h_f.cpu().numpy(), h.detach().numpy(), rtol=1e-4, atol=2e-5)
    assert np.allclose(c_f.cpu().numpy(), c.detach().numpy(), rtol=1e-4, atol=2e-5)
    opt.begin()
    gx = m.replay_backward(ctx, t(gy), 0.0)
    grads = torch.autograd.grad((want * torch.tensor(gy, dtype=D)).sum(),
                                [x64, om.w_i, om.w_h, om.b_h])
    s = max(1.0, float(np.sqrt(T * B)))
    assert np.allclose(gx.cpu().numpy(), grads[0].numpy(), rtol=1e-3, atol=1e-4)
    for p, w in zip((m.w_i, m.w_h, m.b_h), grads[1:]):
        assert np.allclose(p.grad.cpu().numpy(), w.numpy(), rtol=1e-3, atol=3e-5 * s)


def test_lstm_contract(dev):
    m = _lstm(16, 32).to(dev)
    st = m.initialize_state(8)
    assert isinstance(st, tuple) and len(st) == 2
    assert all(s.shape == (8, 32) and float(s.abs().sum()) == 0 for s in st)   # zeros init
    x = torch.ones(8, 16, device=dev)
    o = m(st, x)
    assert o.output.shape == (8, 32) and torch.equal(o.output, o.next_state[0])
    assert o.regularization_loss.shape == (8,) and float(o.regularization_loss.abs().sum()) == 0
    assert o.rollout_extras is None and o.metrics == {}
    rs = m.reset_state(o.next_state)
    assert all(r.shape == (8, 32) and float(r.abs().sum()) == 0 for r in rs)  # zeros-like reset
    # minibatch slicing == full batch (recurrent_test.py:184-211)
    x2 = torch.ones(8, 16, device=dev) * 2
    full = m(o.next_state, x2).output
    sl = lambda s, a, b: tuple(v[a:b].contiguous() for v in s)
    a = m(sl(o.next_state, 0, 4), x2[:4].contiguous()).output
    b = m(sl(o.next_state, 4, 8), x2[4:].contiguous()).output
    assert torch.equal(full[:4], a) and torch.equal(full[4:], b)
    # step-wise calls == sequence replay without resets
    xs = torch.randn(5, 8, 16, device=dev)
    s = st
    outs = []
    for k in range(5):
        r = m(s, xs[k])
        s = r.next_state
        outs.append(r.output)
    _, seq, _, sf = m.replay(st, xs, torch.zeros(5, 8, dtype=torch.bool, device=dev), None, False)
    assert torch.equal(seq, torch.stack(outs))
    assert torch.equal(sf[0], s[0]) and torch.equal(sf[1], s[1])
    with pytest.raises(NotImplementedError):  # swish keeps no pre-activation for BPTT
        from nnx_ppo_amd.networks import activations
        from nnx_ppo_amd.networks.recurrent import LSTM
        from nnx_ppo_amd.networks.types import Rngs

        LSTM(4, 8, Rngs(0), gate_fn=activations.swish)


def _lstm_actor_critic(obs, act, H, critic_h, rngs):
    """The hand-composed LSTM actor of recurrent_test.py:245-261."""
    from nnx_ppo_amd.networks import activations, factories
    from nnx_ppo_amd.networks.adapter import PPOAdapter
    from nnx_ppo_amd.networks.containers import Sequential
    from nnx_ppo_amd.networks.feedforward import Dense
    from nnx_ppo_amd.networks.normalizer import Normalizer
    from nnx_ppo_amd.networks.recurrent import LSTM
    from nnx_ppo_amd.networks.sampling_layers import NormalTanhSampler

    actor = Sequential([Dense(obs, H, rngs, activation=activations.relu), LSTM(H, H, rngs),
                        Dense(H, 2 * act, rngs, activation=None),
                        NormalTanhSampler(rngs, entropy_weight=1e-3)])
    critic = factories.make_mlp([obs] + critic_h + [1], rngs, activation_last_layer=False)
    return Sequential([Normalizer(obs), PPOAdapter(action=actor, value=critic)])


def test_ppo_step_with_lstm_vs_oracle(dev):
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks.types import Rngs

    N, T = 64, 20
    net = _lstm_actor_critic(16, 4, 32, [32], Rngs(42))
    env, oenv = MockEnv(16, 4, max_steps=5), MockEnv(16, 4, max_steps=5)
    ts = ppo.new_training_state(env, net, N, 42, 1e-4, 1.0, device=dev)
    onet = on.from_product(net)
    ots = op.new_training_state(oenv, onet, N, 42, keys, 1e-4, 1.0)
    for k in range(2):
        ts, m = ppo.ppo_step(env, ts, N, T, 0.95, 0.99, 0.2, True, False, 2, 2)
        ots, info = op.ppo_step(oenv, ots, N, T, 0.95, 0.99, 0.2, True, 2, 2, keys)
        assert int(info["rollout"].done.sum()) >= 3 * N  # resets every 5 steps
        for name in ("actor", "critic", "regularization"):
            got, want = m[f"losses/{name}/mean"].item(), info[name].numpy().mean()
            assert np.isfinite(got) and np.allclose(got, want, rtol=2e-3, atol=1e-5), (k, name)
        carry = ts.network_states[1]["action"][1]
        ocarry = ots.network_states[1]["action"][1]
        for a, b in zip(carry, ocarry):
            assert np.allclose(a.cpu().numpy(), b.numpy(), atol=1e-4)
    for p, q in zip(net.parameters(), onet.parameters()):
        assert torch.isfinite(p.data).all()
        assert float((p.data.cpu() - q.detach()).abs().max()) < 5e-4


def test_lstm_graph_capture(dev):
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.graph import GraphedPPOStep
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks.types import Rngs

    N, T = 64, 10
    net = _lstm_actor_critic(16, 4, 32, [32], Rngs(7))
    env = MockEnv(16, 4, max_steps=5)
    ts = ppo.new_training_state(env, net, N, 7, 1e-4, device=dev)
    step = GraphedPPOStep(env, ts, N, T, 0.95, 0.99, 0.2, True, False, 2, 2, warmup=1)
    for _ in range(3):
        ts2, m = step()
    assert int(ts2.steps_taken) == 4 * N * T
    assert all(np.isfinite(float(v)) for k, v in m.items() if k.startswith("losses/"))


@pytest.mark.parametrize("T,B,I,H", [(1, 1, 3, 32), (30, 64, 64, 64), (12, 37, 5, 64),
                                     (5, 33, 8, 128), (6, 20, 7, 96), (3, 4099, 5, 64)])
def test_lstm_matrix_core_path_vs_oracle(dev, T, B, I, H):
    """bf16 compute: h W_h and d_gates W_h^T on the matrix cores (operands rounded to bf16,
    fp32 accumulation, fp32 cell and carries) — against the fp64 oracle with the bf16
    bound of the Dense layers."""
    from nnx_ppo_amd import config
    from nnx_ppo_amd.optim import Optimizer

    prev = config.compute_dtype()
    config.set_compute_dtype("bf16")
    try:
        m = _lstm(I, H, seed=T + B)
        m.to(dev)
        assert m._mfma()
        opt = Optimizer(m, 1e-3, device=dev)
        om = on.from_product(m)
        rng = np.random.default_rng(B)
        x = rng.normal(size=(T, B, I)).astype(np.float32)
        h0 = rng.normal(size=(B, H)).astype(np.float32)
        c0 = rng.normal(size=(B, H)).astype(np.float32)
        done = rng.random((T, B)) < 0.2
        gy = rng.normal(size=(T, B, H)).astype(np.float32)
        t = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt).to(dev)
        ctx, out, _, (h_f, c_f) = m.replay((t(h0), t(c0)), t(x), t(done, torch.bool), None, True)
        x64 = torch.tensor(x, dtype=D, requires_grad=True)
        h, c = torch.tensor(h0, dtype=D), torch.tensor(c0, dtype=D)
        outs = []
        for k in range(T):
            o = om((h, c), x64[k])
            outs.append(o.output)
            d = torch.tensor(done[k])[:, None]
            h = torch.where(d, torch.zeros_like(o.next_state[0]), o.next_state[0])
            c = torch.where(d, torch.zeros_like(o.next_state[1]), o.next_state[1])
        want = torch.stack(outs)
        assert np.allclose(out.cpu().numpy(), want.detach().numpy(), atol=3e-2)
        assert np.allclose(h_f.cpu().numpy(), h.detach().numpy(), atol=3e-2)
        assert np.allclose(c_f.cpu().numpy(), c.detach().numpy(), atol=5e-2)
        # single-step calls == sequence replay (same kernel, bit for bit)
        st = (t(h0), t(c0))
        for k in range(min(T, 3)):
            r = m(st, t(x[k]))
            assert torch.equal(r.output, out[k])
            dk = t(done[k], torch.bool)[:, None]
            st = tuple(torch.where(dk, torch.zeros_like(s), s) for s in r.next_state)
        opt.begin()
        gx = m.replay_backward(ctx, t(gy), 0.0)
        grads = torch.autograd.grad((want * torch.tensor(gy, dtype=D)).sum(),
                                    [x64, om.w_i, om.w_h, om.b_h])

        def close(a, b):
            a, b = a.cpu().numpy().ravel(), b.numpy().ravel()
            cos = float(a @ b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
            rel = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
            return cos > 0.995 and rel < 0.08

        assert close(gx, grads[0])
        for p_, w in zip((m.w_i, m.w_h, m.b_h), grads[1:]):
            assert close(p_.grad, w)
    finally:
        config.set_compute_dtype(prev)


# ---- trainable initial state, gate functions, wide cells (recurrent.py:31-90,132-161) ------
def _opt_lstm(dev, in_f, H, seed=0, **kw):
    from nnx_ppo_amd.networks.recurrent import LSTM
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.optim import Optimizer

    m = LSTM(in_f, H, Rngs(seed), **kw)
    rng = np.random.default_rng(seed + 1)
    m.b_h.data = torch.tensor(rng.normal(0, 0.1, size=4 * H), dtype=torch.float32)
    if kw.get("trainable_initial_state"):
        m.initial_h.data = torch.tensor(rng.normal(0, 0.5, size=H), dtype=torch.float32)
        m.initial_c.data = torch.tensor(rng.normal(0, 0.5, size=H), dtype=torch.float32)
    m.to(dev)
    opt = Optimizer(m, 1e-3, device=dev)
    return m, opt


def test_lstm_trainable_initial_state_contract(dev):
    """recurrent_test.py:95-152: the initial state is a pair of learnable [H] vectors
    (zeros at construction), broadcast to the batch by `initialize_state` and to the
    previous state's shape by `reset_state`."""
    from nnx_ppo_amd.networks.recurrent import LSTM
    from nnx_ppo_amd.networks.types import Rngs

    m = LSTM(4, 8, Rngs(0), trainable_initial_state=True).to(dev)
    names = [n for n, _ in m.named_parameters()]
    assert "initial_h" in names and "initial_c" in names
    assert m.initial_h.shape == (8,) and float(m.initial_h.data.abs().sum()) == 0
    st = m.initialize_state(3)
    assert all(s.shape == (3, 8) and float(s.abs().sum()) == 0 for s in st)
    m.initial_h.data.copy_(torch.arange(8.0))
    m.initial_c.data.copy_(-torch.arange(8.0))
    h, c = m.initialize_state(5)
    assert h.shape == (5, 8) and torch.equal(h[3].cpu(), torch.arange(8.0))
    assert torch.equal(c[0].cpu(), -torch.arange(8.0))
    out = m((h, c), torch.ones(5, 4, device=dev))
    rh, rc = m.reset_state(out.next_state)
    assert torch.equal(rh, h) and torch.equal(rc, c)
    # without the option there are no such parameters
    assert "initial_h" not in [n for n, _ in LSTM(4, 8, Rngs(0)).named_parameters()]


@pytest.mark.parametrize("T,B,I,H", [(9, 37, 5, 32), (30, 64, 16, 64), (6, 2051, 5, 64),
                                     (5, 19, 7, 320)])
@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_lstm_trainable_init_replay_vs_oracle(dev, T, B, I, H, compute):
    """Replay with resets to the LEARNED initial state, and BPTT including d initial_h /
    d initial_c (what flows into the carry of every step that follows a done), against
    fp64 autograd through the oracle's step-wise scan (ppo.py:411-418).  Same numbers under
    compute_dtype bf16: this option runs the fp32 recurrence kernel either way."""
    from nnx_ppo_amd import config

    with config.use_compute_dtype(compute):
        m, opt = _opt_lstm(dev, I, H, seed=T + B, trainable_initial_state=True)
        om = on.from_product(m)
        rng = np.random.default_rng(B)
        x = rng.normal(size=(T, B, I)).astype(np.float32)
        h0 = rng.normal(size=(B, H)).astype(np.float32)
        c0 = rng.normal(size=(B, H)).astype(np.float32)
        done = rng.random((T, B)) < 0.25
        gy = rng.normal(size=(T, B, H)).astype(np.float32)
        t = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt).to(dev)
        ctx, out, reg, (h_f, c_f) = m.replay((t(h0), t(c0)), t(x), t(done, torch.bool), None,
                                              need_input_grad=True)
        x64 = torch.tensor(x, dtype=D, requires_grad=True)
        state = (torch.tensor(h0, dtype=D), torch.tensor(c0, dtype=D))
        outs = []
        for k in range(T):
            o = om(state, x64[k])
            outs.append(o.output)
            state = op.tree_where(torch.tensor(done[k]), om.reset_state(o.next_state),
                                  o.next_state)
        want = torch.stack(outs)
        assert np.allclose(out.cpu().numpy(), want.detach().numpy(), rtol=1e-4, atol=3e-5)
        assert np.allclose(h_f.cpu().numpy(), state[0].detach().numpy(), rtol=1e-4, atol=3e-5)
        assert np.allclose(c_f.cpu().numpy(), state[1].detach().numpy(), rtol=1e-4, atol=3e-5)
        opt.begin()
        gx = m.replay_backward(ctx, t(gy), 0.0)
        grads = torch.autograd.grad((want * torch.tensor(gy, dtype=D)).sum(),
                                    [x64, om.w_i, om.w_h, om.b_h, om.initial_h, om.initial_c])
        s = max(1.0, float(np.sqrt(T * B)))
        assert np.allclose(gx.cpu().numpy(), grads[0].numpy(), rtol=1e-3, atol=1e-4)
        for p, w in zip((m.w_i, m.w_h, m.b_h, m.initial_h, m.initial_c), grads[1:]):
            assert np.allclose(p.grad.cpu().numpy(), w.numpy(), rtol=1e-3, atol=3e-5 * s)
        assert float(m.initial_h.grad.abs().sum()) > 0  # there were resets to learn from


@pytest.mark.parametrize("gate,act", [("sigmoid", "relu"), ("tanh", "tanh"),
                                      ("sigmoid", "identity"), ("relu", "sigmoid")])
def test_lstm_gate_functions_vs_oracle(dev, gate, act):
    """`gate_fn` / `activation_fn` (recurrent.py:36-37): forward and every gradient."""
    T, B, I, H = 8, 33, 6, 48
    m, opt = _opt_lstm(dev, I, H, seed=3, gate_fn=gate, activation_fn=act)
    m.w_h.data.mul_(0.5)  # keep the unbounded (relu / identity) cells in range
    om = on.from_product(m)
    rng = np.random.default_rng(5)
    x = rng.normal(size=(T, B, I)).astype(np.float32)
    done = rng.random((T, B)) < 0.2
    gy = rng.normal(size=(T, B, H)).astype(np.float32)
    t = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt).to(dev)
    st = m.initialize_state(B)
    ctx, out, _, _ = m.replay(st, t(x), t(done, torch.bool), None, need_input_grad=True)
    x64 = torch.tensor(x, dtype=D, requires_grad=True)
    state = om.initialize_state(B)
    outs = []
    for k in range(T):
        o = om(state, x64[k])
        outs.append(o.output)
        state = op.tree_where(torch.tensor(done[k]), om.reset_state(o.next_state), o.next_state)
    want = torch.stack(outs)
    assert np.allclose(out.cpu().numpy(), want.detach().numpy(), rtol=1e-4, atol=3e-5)
    opt.begin()
    gx = m.replay_backward(ctx, t(gy), 0.0)
    grads = torch.autograd.grad((want * torch.tensor(gy, dtype=D)).sum(),
                                [x64, om.w_i, om.w_h, om.b_h])
    s = float(np.sqrt(T * B))
    assert np.allclose(gx.cpu().numpy(), grads[0].numpy(), rtol=1e-3, atol=1e-4)
    for p, w in zip((m.w_i, m.w_h, m.b_h), grads[1:]):
        assert np.allclose(p.grad.cpu().numpy(), w.numpy(), rtol=1e-3, atol=3e-5 * s)
    # the one-step interface agrees with the sequence kernel
    o1 = m(st, t(x[0]))
    assert torch.equal(o1.output, out[0])


def test_ppo_step_with_trainable_initial_state_vs_oracle(dev):
    """The whole iteration with an LSTM actor whose reset state is learned: rollout resets
    use it (rollout.py:42-43), the loss replay resets to it and trains it."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks import activations, factories
    from nnx_ppo_amd.networks.adapter import PPOAdapter
    from nnx_ppo_amd.networks.containers import Sequential
    from nnx_ppo_amd.networks.feedforward import Dense
    from nnx_ppo_amd.networks.normalizer import Normalizer
    from nnx_ppo_amd.networks.recurrent import LSTM
    from nnx_ppo_amd.networks.sampling_layers import NormalTanhSampler
    from nnx_ppo_amd.networks.types import Rngs

    rngs = Rngs(9)
    lstm = LSTM(32, 32, rngs, trainable_initial_state=True)
    actor = Sequential([Dense(16, 32, rngs, activation=activations.relu), lstm,
                        Dense(32, 8, rngs, activation=None),
                        NormalTanhSampler(rngs, entropy_weight=1e-3)])
    critic = factories.make_mlp([16, 32, 1], rngs, activation_last_layer=False)
    net = Sequential([Normalizer(16), PPOAdapter(action=actor, value=critic)])
    gen = np.random.default_rng(1)
    lstm.initial_h.data = torch.tensor(gen.normal(0, 0.3, 32), dtype=torch.float32)
    lstm.initial_c.data = torch.tensor(gen.normal(0, 0.3, 32), dtype=torch.float32)
    N, T = 64, 16
    env, oenv = MockEnv(16, 4, max_steps=5), MockEnv(16, 4, max_steps=5)
    # lr 1e-4 with clipping, as the sibling test: Adam's first steps move every parameter by
    # ~lr whatever its gradient's size, so larger steps amplify fp32-vs-fp64 noise in
    # near-zero gradients into the later gradient steps' losses
    ts = ppo.new_training_state(env, net, N, 42, 1e-4, 1.0, device=dev)
    onet = on.from_product(net)
    ots = op.new_training_state(oenv, onet, N, 42, keys, 1e-4, 1.0)
    ih0 = lstm.initial_h.data.clone()
    for k in range(2):
        ts, m = ppo.ppo_step(env, ts, N, T, 0.95, 0.99, 0.2, True, False, 2, 2)
        ots, info = op.ppo_step(oenv, ots, N, T, 0.95, 0.99, 0.2, True, 2, 2, keys)
        for name in ("actor", "critic", "regularization"):
            got, want = m[f"losses/{name}/mean"].item(), info[name].numpy().mean()
            assert np.isfinite(got) and np.allclose(got, want, rtol=2e-3, atol=1e-5), (k, name)
    assert not torch.equal(lstm.initial_h.data, ih0)  # it is being trained
    for (n, p), q in zip(net.named_parameters(), onet.parameters()):
        assert float((p.data.cpu() - q.detach()).abs().max()) < 5e-4, n
