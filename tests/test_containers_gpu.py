"""Routing containers composed with Dense layers on the GPU: sequence replay and its
hand-written backward against the oracle twins' fp64 autograd (fp32 compute path:
1e-4 rel on outputs, 1e-3 rel on gradients)."""
import numpy as np
import pytest
import torch

from oracle import networks as on

pytestmark = pytest.mark.gpu
D = torch.float64


def _build(name, rngs):
    from nnx_ppo_amd.networks import activations as A
    from nnx_ppo_amd.networks.containers import Concat, Parallel, Sequential, Splitter
    from nnx_ppo_amd.networks.feedforward import Dense
    from nnx_ppo_amd.networks.utils import Map, Merge, Scale

    d = lambda i, o, act=None: Dense(i, o, rngs, activation=act)
    if name == "concat":
        return Sequential([Concat(a=d(3, 4, A.relu), b=d(5, 2, A.tanh)), d(6, 3)]), {"a": 3, "b": 5}
    if name == "parallel":
        return Sequential([d(4, 8, A.relu),
                           Parallel(p=d(8, 3), q=Sequential([d(8, 5, A.tanh), Scale(0.5)]))]), 4
    if name == "splitter_map":
        return Sequential([d(4, 7), Splitter(u=3, v=2), Map(u=d(3, 2, A.swish), v=d(2, 2))]), 4
    if name == "merge":
        return Merge(m1=Sequential([d(4, 6, A.relu), Splitter(a=2, b=4)]),
                     m2=Sequential([d(4, 3), Splitter(c=3)])), 4
    raise KeyError(name)


def _leaves(t):
    from nnx_ppo_amd.tree import tree_leaves

    return tree_leaves(t)


@pytest.mark.parametrize("name", ["concat", "parallel", "splitter_map", "merge"])
def test_container_replay_and_backward_vs_oracle(dev, name):
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.optim import Optimizer
    from nnx_ppo_amd.tree import tree_map

    T, B = 3, 17
    net, in_spec = _build(name, Rngs(5))
    net.to(dev)
    opt = Optimizer(net, 1e-3, device=dev)
    onet = on.from_product(net)
    rng = np.random.default_rng(3)
    mk = lambda f: rng.normal(size=(T, B, f)).astype(np.float32)
    x = {k: mk(f) for k, f in in_spec.items()} if isinstance(in_spec, dict) else mk(in_spec)
    t = lambda a: torch.as_tensor(a).to(dev)
    x_dev = tree_map(t, x)
    state = net.initialize_state(B)
    done = torch.zeros(T, B, dtype=torch.bool, device=dev)
    ctx, out, reg, _ = net.replay(state, x_dev, done, None, need_input_grad=True)
    # oracle on the flattened [T*B, F] batch (the modules are stateless)
    x64 = tree_map(lambda a: torch.tensor(a.reshape(T * B, -1), dtype=D, requires_grad=True), x)
    want = onet(onet.initialize_state(T * B), x64).output
    g = tree_map(lambda w: rng.normal(size=tuple(w.shape)).astype(np.float32), want)
    for a, b in zip(_leaves(out), _leaves(want)):
        assert np.allclose(a.reshape(T * B, -1).cpu().numpy(), b.detach().numpy(),
                           rtol=1e-4, atol=1e-5)
    obj = sum((w * torch.tensor(gi, dtype=D)).sum() for w, gi in zip(_leaves(want), _leaves(g)))
    grads = torch.autograd.grad(obj, _leaves(x64) + onet.parameters())
    opt.begin()
    g_dev = tree_map(lambda gi, o: t(gi).reshape(o.shape), g, out)
    gx = net.replay_backward(ctx, g_dev, 0.0)
    n_in = len(_leaves(x64))
    for a, b in zip(_leaves(gx), grads[:n_in]):
        assert np.allclose(a.reshape(T * B, -1).cpu().numpy(), b.numpy(), rtol=1e-3, atol=1e-5)
    for p, w in zip(net.parameters(), grads[n_in:]):
        assert np.allclose(p.grad.cpu().numpy(), w.numpy(), rtol=1e-3, atol=1e-5), name


def test_single_step_call_matches_replay(dev):
    from nnx_ppo_amd.networks.types import Rngs

    net, _ = _build("parallel", Rngs(9))
    net.to(dev)
    x = torch.randn(6, 11, 4, device=dev)
    st = net.initialize_state(11)
    _, seq, _, _ = net.replay(st, x, torch.zeros(6, 11, dtype=torch.bool, device=dev), None, False)
    for k in range(6):
        o = net(st, x[k]).output
        for key in ("p", "q"):
            assert torch.allclose(o[key], seq[key][k], rtol=1e-6, atol=1e-6)
