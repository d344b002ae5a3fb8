"""GRU carry (BASELINE config 4).  The reference has no GRU: the module
contract is its LSTM wrapper's (recurrent_test.py:59-93 zeros init/reset,
184-211 minibatch slicing == full batch, 213-230 regularisation shape,
232-343 rollout / ppo_step with resets every 5 steps), the cell arithmetic is
flax's GRUCell (PARITY UNPINNED by the reference; pinned here only against the
oracle's restatement and fp64 autograd)."""
import numpy as np
import pytest
import torch

from nnx_ppo_amd import random as keys
from oracle import networks as on
from oracle import ppo as op

pytestmark = pytest.mark.gpu
D = torch.float64


def _gru(in_f, H, seed=0):
    from nnx_ppo_amd.networks.recurrent import GRU
    from nnx_ppo_amd.networks.types import Rngs

    g = GRU(in_f, H, Rngs(seed))
    rng = np.random.default_rng(seed + 1)
    g.b_i.data = torch.tensor(rng.normal(0, 0.1, size=3 * H), dtype=torch.float32)
    g.b_hn.data = torch.tensor(rng.normal(0, 0.1, size=H), dtype=torch.float32)
    return g


@pytest.mark.parametrize("T,B,I,H", [(1, 1, 3, 16), (30, 64, 64, 64), (12, 37, 5, 64),
                                     (7, 100, 16, 32), (5, 33, 8, 128), (6, 20, 7, 80),
                                     # 8 and 16 envs per workgroup (B >= 2048 / 4096)
                                     (4, 2051, 5, 64), (3, 4099, 5, 64)])
def test_gru_sequence_fwd_bwd_vs_oracle(dev, T, B, I, H):
    from nnx_ppo_amd.optim import Optimizer

    g = _gru(I, H, seed=T + B)
    g.to(dev)
    opt = Optimizer(g, 1e-3, device=dev)
    og = on.from_product(g)
    rng = np.random.default_rng(B)
    x = rng.normal(size=(T, B, I)).astype(np.float32)
    h0 = rng.normal(size=(B, H)).astype(np.float32)
    done = rng.random((T, B)) < 0.2
    gy = rng.normal(size=(T, B, H)).astype(np.float32)
    t = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt).to(dev)
    ctx, out, reg, h_final = g.replay(t(h0), t(x), t(done, torch.bool), None, need_input_grad=True)
    assert reg is None and out.shape == (T, B, H)
    # oracle: step-wise scan with reset-on-done (ppo.py:411-418)
    x64 = torch.tensor(x, dtype=D, requires_grad=True)
    h = torch.tensor(h0, dtype=D)
    outs = []
    for k in range(T):
        o = og(h, x64[k])
        outs.append(o.output)
        h = torch.where(torch.tensor(done[k])[:, None], torch.zeros_like(o.next_state), o.next_state)
    want = torch.stack(outs)
    assert np.allclose(out.cpu().numpy(), want.detach().numpy(), rtol=1e-4, atol=2e-5)
    assert np.allclose(h_final.cpu().numpy(), h.detach().numpy(), rtol=1e-4, atol=2e-5)
    opt.begin()
    gx = g.replay_backward(ctx, t(gy), 0.0)
    grads = torch.autograd.grad((want * torch.tensor(gy, dtype=D)).sum(),
                                [x64, og.w_i, og.b_i, og.w_h, og.b_hn])
    s = max(1.0, float(np.sqrt(T * B)))
    assert np.allclose(gx.cpu().numpy(), grads[0].numpy(), rtol=1e-3, atol=1e-4)
    for p, w in zip((g.w_i, g.b_i, g.w_h, g.b_hn), grads[1:]):
        assert np.allclose(p.grad.cpu().numpy(), w.numpy(), rtol=1e-3, atol=3e-5 * s)


def test_gru_contract(dev):
    g = _gru(16, 32).to(dev)
    st = g.initialize_state(8)
    assert st.shape == (8, 32) and float(st.abs().sum()) == 0          # zeros init
    x = torch.ones(8, 16, device=dev)
    o = g(st, x)
    assert o.output.shape == (8, 32) and torch.equal(o.output, o.next_state)
    assert o.regularization_loss.shape == (8,) and float(o.regularization_loss.abs().sum()) == 0
    assert o.rollout_extras is None
    assert float(g.reset_state(o.next_state).abs().sum()) == 0         # zeros-like reset
    # minibatch slicing == full batch (recurrent_test.py:184-211)
    x2 = torch.ones(8, 16, device=dev) * 2
    full = g(o.next_state, x2).output
    a = g(o.next_state[:4].contiguous(), x2[:4].contiguous()).output
    b = g(o.next_state[4:].contiguous(), x2[4:].contiguous()).output
    assert torch.equal(full[:4], a) and torch.equal(full[4:], b)
    # step-wise calls == sequence replay without resets
    xs = torch.randn(5, 8, 16, device=dev)
    h = st
    outs = []
    for k in range(5):
        r = g(h, xs[k])
        h = r.next_state
        outs.append(r.output)
    _, seq, _, hf = g.replay(st, xs, torch.zeros(5, 8, dtype=torch.bool, device=dev), None, False)
    assert torch.equal(seq, torch.stack(outs)) and torch.equal(hf, h)


def test_ppo_step_with_gru_vs_oracle(dev):
    """recurrent_test.py:232-343 shape: env resets every 5 steps, gradient
    clipping 1.0; plus full parity of the iteration against the oracle."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    N, T = 64, 20
    net = factories.make_gru_actor_critic(16, 4, 32, [32], Rngs(42), entropy_weight=1e-3)
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
        assert np.allclose(carry.cpu().numpy(), ocarry.numpy(), atol=1e-4)
    for p, q in zip(net.parameters(), onet.parameters()):
        assert torch.isfinite(p.data).all()
        assert float((p.data.cpu() - q.detach()).abs().max()) < 5e-4


def test_gru_graph_capture(dev):
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.graph import GraphedPPOStep
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    def setup():
        env = MockEnv(5, 1, max_steps=5)
        net = factories.make_gru_actor_critic(5, 1, 64, [64], Rngs(1))
        return env, net, ppo.new_training_state(env, net, 128, 2, 1e-3, device=dev)

    args = (128, 8, 0.95, 0.99, 0.2, True, False, 2, 2)
    ea, na, ta = setup()
    for _ in range(3):
        ta, ma = ppo.ppo_step(ea, ta, *args)
    eb, nb, tb = setup()
    g = GraphedPPOStep(eb, tb, *args, warmup=1)
    for _ in range(2):
        tb, mb = g()
    torch.cuda.synchronize()
    for pa, pb in zip(na.parameters(), nb.parameters()):
        assert torch.equal(pa.data, pb.data)
    assert torch.equal(ta.network_states[1]["action"][1], tb.network_states[1]["action"][1])


@pytest.mark.parametrize("T,B,I,H", [(1, 1, 3, 32), (30, 64, 64, 64), (12, 37, 5, 64),
                                     (5, 33, 8, 128), (6, 20, 7, 96), (3, 4099, 5, 64)])
def test_gru_matrix_core_path_vs_oracle(dev, T, B, I, H):
    """bf16 compute: h W_h and dgh W_h^T on the matrix cores (operands rounded to bf16,
    fp32 accumulation, fp32 cell) — against the fp64 oracle with the bf16 bound of the
    Dense layers (2e-2 abs on O(1) activations, 5 % on gradients)."""
    from nnx_ppo_amd import config
    from nnx_ppo_amd.optim import Optimizer

    prev = config.compute_dtype()
    config.set_compute_dtype("bf16")
    try:
        g = _gru(I, H, seed=T + B)
        g.to(dev)
        assert g._mfma()
        opt = Optimizer(g, 1e-3, device=dev)
        og = on.from_product(g)
        rng = np.random.default_rng(B)
        x = rng.normal(size=(T, B, I)).astype(np.float32)
        h0 = rng.normal(size=(B, H)).astype(np.float32)
        done = rng.random((T, B)) < 0.2
        gy = rng.normal(size=(T, B, H)).astype(np.float32)
        t = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt).to(dev)
        ctx, out, _, h_final = g.replay(t(h0), t(x), t(done, torch.bool), None, True)
        x64 = torch.tensor(x, dtype=D, requires_grad=True)
        h = torch.tensor(h0, dtype=D)
        outs = []
        for k in range(T):
            o = og(h, x64[k])
            outs.append(o.output)
            h = torch.where(torch.tensor(done[k])[:, None], torch.zeros_like(o.next_state),
                            o.next_state)
        want = torch.stack(outs)
        assert np.allclose(out.cpu().numpy(), want.detach().numpy(), atol=2e-2)
        assert np.allclose(h_final.cpu().numpy(), h.detach().numpy(), atol=2e-2)
        # single-step calls == sequence replay (same kernel, bit for bit)
        hs = t(h0)
        for k in range(min(T, 3)):
            r = g(hs, t(x[k]))
            assert torch.equal(r.output, out[k])
            hs = torch.where(t(done[k], torch.bool)[:, None], torch.zeros_like(r.next_state),
                             r.next_state)
        opt.begin()
        gx = g.replay_backward(ctx, t(gy), 0.0)
        grads = torch.autograd.grad((want * torch.tensor(gy, dtype=D)).sum(),
                                    [x64, og.w_i, og.b_i, og.w_h, og.b_hn])
        def close(a, b):
            a, b = a.cpu().numpy().ravel(), b.numpy().ravel()
            cos = float(a @ b) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
            rel = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
            return cos > 0.995 and rel < 0.08
        assert close(gx, grads[0])
        for p_, w in zip((g.w_i, g.b_i, g.w_h, g.b_hn), grads[1:]):
            assert close(p_.grad, w)
    finally:
        config.set_compute_dtype(prev)


@pytest.mark.parametrize("T,B,H", [(30, 1024, 64), (7, 37, 96), (3, 4099, 128)])
def test_gru_matrix_core_rows_per_workgroup(dev, monkeypatch, T, B, H):
    """`rows_per_group` (csrc/gru_mfma.hip): a small batch is spread over more workgroups by
    filling only 8 or 4 rows of each 16-row tile.  Rows do not interact, so the sequence kernels
    — forward (training form) and BPTT — must return the same bits whatever the fill."""
    from nnx_ppo_amd import ops

    gen = torch.Generator(device="cpu").manual_seed(T * B + H)
    r = lambda *s: torch.randn(*s, generator=gen).to(dev)
    gi, w_h, b, h0, g_h = r(T, B, 3 * H), r(H, 3 * H) / H ** 0.5, r(H), r(B, H), r(T, B, H)
    done = (torch.rand(T, B, generator=gen) < 0.2).to(dev)
    outs = []
    for rows in ("16", "8", "4", ""):
        monkeypatch.setenv("MIPPO_GRU_ROWS", rows)
        h_out, h_prev, gates, h_final = ops.gru_seq_fwd(gi, w_h, b, h0, done, True, True)
        dgi, dgh = ops.gru_seq_bwd(g_h, gates, h_prev, w_h, done, mfma=True, dgh_as_bf16=True)
        outs.append((h_out, h_prev, h_prev.bf16_image, gates, h_final, dgi, dgh))
    for other in outs[1:]:
        for a, b_ in zip(outs[0], other):
            assert torch.equal(a, b_)


@pytest.mark.parametrize("B,H,rows", [(8, 32, ""), (40, 64, "16"), (4100, 64, "")])
def test_gru_matrix_core_long_sequence_equals_chunks(dev, monkeypatch, B, H, rows):
    """Sequences longer than the kernels' 240-step window of done flags (`DoneWindow`) and not a
    multiple of the look-ahead: one T = 611 call == the same steps in chunks of 97 with the carry
    (forward) and the carry's gradient (BPTT) handed from chunk to chunk by the caller — exact,
    because a step only sees its carry.  Covers the restaging of the window in both directions,
    in the 4-row and the full-tile form."""
    from nnx_ppo_amd import ops

    monkeypatch.setenv("MIPPO_GRU_ROWS", rows)
    T, C = 611, 97
    gen = torch.Generator(device="cpu").manual_seed(B + H)
    r = lambda *s: torch.randn(*s, generator=gen).to(dev)
    gi, w_h, b, h0, g_h = r(T, B, 3 * H), r(H, 3 * H) / H ** 0.5, r(H), r(B, H), r(T, B, H) * 0.1
    done = (torch.rand(T, B, generator=gen) < 0.05).to(dev)
    h_out, h_prev, gates, h_final = ops.gru_seq_fwd(gi, w_h, b, h0, done, True, True)
    dh0 = torch.empty(B, H, device=dev)
    dgi, dgh = ops.gru_seq_bwd(g_h, gates, h_prev, w_h, done, mfma=True, dh0_out=dh0)
    # forward in chunks
    h, outs = h0, []
    for t0 in range(0, T, C):
        o = ops.gru_seq_fwd(gi[t0:t0 + C].contiguous(), w_h, b, h, done[t0:t0 + C].contiguous(),
                            True, True)
        outs.append(o)
        h = o[3]
    assert torch.equal(torch.cat([o[0] for o in outs]), h_out)
    assert torch.equal(torch.cat([o[1] for o in outs]), h_prev)
    assert torch.equal(torch.cat([o[2] for o in outs]), gates)
    assert torch.equal(h, h_final)
    # BPTT in chunks, last chunk first: d loss / d carry enters the chunk before through its
    # last step's output gradient (masked where that step ended an episode)
    carry, pieces = None, []
    starts = list(range(0, T, C))
    for t0 in reversed(starts):
        g = g_h[t0:t0 + C].clone()
        if carry is not None:
            g[-1] += torch.where(done[t0 + C - 1][:, None], torch.zeros_like(carry), carry)
        d0 = torch.empty(B, H, device=dev)
        pieces.append(ops.gru_seq_bwd(g, gates[t0:t0 + C].contiguous(),
                                      h_prev[t0:t0 + C].contiguous(), w_h,
                                      done[t0:t0 + C].contiguous(), mfma=True, dh0_out=d0))
        carry = d0
    pieces.reverse()
    assert torch.equal(torch.cat([p[0] for p in pieces]), dgi)
    assert torch.equal(torch.cat([p[1] for p in pieces]), dgh)
    assert torch.equal(carry, dh0)
