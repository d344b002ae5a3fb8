"""bf16 MFMA path.  Kernel-level: against an fp64 evaluation on the SAME
bf16-rounded operands (so the only admissible difference is fp32 accumulation
order: 1e-4 rel, and the bf16 rounding of bf16 outputs: 2^-8 rel).  End to end:
against the fp64 oracle with the looser stated bound for bf16 operands (SURVEY
§8a: ~1e-2 rel on logits) — losses within 5e-2 (actor) / 2e-3 (critic, reg)."""
import math

import numpy as np
import pytest
import torch

from nnx_ppo_amd import random as keys
from oracle import networks as on
from oracle import ppo as op

pytestmark = pytest.mark.gpu
D = torch.float64
BF = torch.bfloat16


def _r(a):  # round to bf16, back to fp64
    return torch.as_tensor(a, dtype=torch.float32).to(BF).to(D)


def _act(z, act):
    return {"none": z, "relu": torch.relu(z), "tanh": torch.tanh(z),
            "swish": z * torch.sigmoid(z)}[act]


@pytest.mark.parametrize("M,K,N", [(1, 1, 1), (4096, 5, 64), (30720, 64, 64), (30720, 64, 2),
                                   (30720, 5, 256), (30720, 256, 256), (30720, 256, 1),
                                   (777, 17, 512), (513, 512, 12), (130, 33, 65), (257, 40, 129)])
@pytest.mark.parametrize("act", ["none", "relu", "tanh", "swish"])
def test_dense_bf16_kernels(dev, M, K, N, act):
    from nnx_ppo_amd import ops

    if act in ("tanh", "swish") and M > 5000:
        pytest.skip("covered by relu/none at this size")
    rng = np.random.default_rng(M + 7 * K + 13 * N)
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(K, N)) / math.sqrt(K)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    code = ops.ACT_CODES[act]
    g = lambda a: torch.as_tensor(a).to(dev)
    x_bf = ops.cast_pad_bf16(g(x))
    assert x_bf.shape == (M, ops.pad8(K))
    assert torch.equal(x_bf[:, :K].cpu(), torch.as_tensor(x).to(BF))
    assert float(x_bf[:, K:].float().abs().sum()) == 0
    w_bf = torch.zeros(K, ops.pad8(N), dtype=BF, device=dev)
    wt_bf = torch.zeros(N, ops.pad8(K), dtype=BF, device=dev)
    ops.weights_to_bf16(g(w), w_bf, wt_bf)
    assert torch.equal(w_bf[:, :N].cpu(), torch.as_tensor(w).to(BF))
    assert torch.equal(wt_bf[:, :K].cpu(), torch.as_tensor(w).to(BF).t())

    y, y_bf, pre = ops.dense_fwd_bf16(x_bf, wt_bf, g(b), K, N, code, want_f32=True,
                                      want_bf=True, want_preact=(act == "swish"))
    z64 = _r(x) @ _r(w) + torch.as_tensor(b, dtype=D)
    y64 = _act(z64, act)
    assert np.allclose(y.cpu().numpy(), y64.numpy(), rtol=1e-4, atol=1e-4)
    assert torch.equal(y_bf[:, :N], y.to(BF))          # bf16 image = rounding of the fp32 output
    assert float(y_bf[:, N:].float().abs().sum()) == 0  # producers write the zero padding
    if act == "swish":
        assert np.allclose(pre[:, :N].float().cpu().numpy(), z64.numpy(), rtol=1e-2, atol=1e-2)
        assert float(pre[:, N:].float().abs().sum()) == 0

    # dX of THIS layer given dz (bf16), times act' of a previous layer's output `prev`
    dz = rng.normal(size=(M, N)).astype(np.float32)
    prev = rng.normal(size=(M, K)).astype(np.float32)
    dz_bf = ops.cast_pad_bf16(g(dz))
    prev_bf = ops.cast_pad_bf16(g(prev))
    gx, gx_bf = ops.dense_bwd_dx_bf16(dz_bf, w_bf, prev_bf, code, K, N, want_f32=True,
                                      want_bf=True)
    p64 = _r(prev)
    dact = {"none": torch.ones_like(p64), "relu": (p64 > 0).to(D), "tanh": 1 - p64 * p64,
            "swish": torch.sigmoid(p64) * (1 + p64 * (1 - torch.sigmoid(p64)))}[act]
    gx64 = (_r(dz) @ _r(w).t()) * dact
    assert np.allclose(gx.cpu().numpy(), gx64.numpy(), rtol=1e-4, atol=1e-4)
    assert torch.equal(gx_bf[:, :K], gx.to(BF))
    assert float(gx_bf[:, K:].float().abs().sum()) == 0

    # dW, db from the row-major operands (transposed on the fly by ds_read_b64_tr_b16)
    gw = torch.zeros(K, N, device=dev)
    gb = torch.zeros(N, device=dev)
    ops.dense_bwd_dw_bf16(x_bf, dz_bf, gw, gb, accumulate=True)
    gw64 = _r(x).t() @ _r(dz)
    gb64 = _r(dz).sum(0)
    s = math.sqrt(M)
    assert np.allclose(gw.cpu().numpy(), gw64.numpy(), rtol=1e-4, atol=2e-5 * s)
    assert np.allclose(gb.cpu().numpy(), gb64.numpy(), rtol=1e-4, atol=2e-5 * s)
    gw2 = torch.zeros(K, N, device=dev)
    ops.dense_bwd_dw_bf16(x_bf, dz_bf, gw2, None, accumulate=False)
    assert torch.equal(gw2, gw)  # bitwise reproducible


def test_dw_asymmetric_layout(dev):
    """x = I (padded) against an asymmetric dz pins the transposing fragment
    gather: gW must equal dz's leading rows exactly, not its transpose."""
    from nnx_ppo_amd import ops

    M, K, N = 192, 70, 40
    x = torch.zeros(M, K, device=dev)
    x[:K, :K] = torch.eye(K, device=dev)
    dz = (torch.arange(M * N, device=dev, dtype=torch.float32).reshape(M, N) % 251) - 125
    gw = torch.zeros(K, N, device=dev)
    ops.dense_bwd_dw_bf16(ops.cast_pad_bf16(x), ops.cast_pad_bf16(dz), gw, None, accumulate=False)
    assert torch.equal(gw, dz[:K].to(BF).float())


def test_cast_pad_with_activation_derivative(dev):
    from nnx_ppo_amd import ops

    M, F = 100, 12
    g = torch.randn(M, F, device=dev)
    y = torch.randn(M, F, device=dev)
    y_bf = ops.cast_pad_bf16(y)
    dz = ops.cast_pad_bf16(g, aux=y_bf, act=ops.ACT_TANH)
    want = (g * (1 - y_bf[:, :F].float() ** 2)).to(BF)
    assert torch.equal(dz[:, :F], want) and float(dz[:, F:].float().abs().sum()) == 0


def _shadows4(ops, w, dev):
    """(row-major W, row-major W^T, forward fragment image, backward fragment image)."""
    K, N = w.shape
    w_bf = torch.zeros(K, ops.pad8(N), dtype=BF, device=dev)
    wt_bf = torch.zeros(N, ops.pad8(K), dtype=BF, device=dev)
    nf, nb = ops.frag_sizes(K, N)
    ff = torch.zeros(nf, dtype=BF, device=dev)
    fb = torch.zeros(nb, dtype=BF, device=dev)
    ops.weights_to_bf16_multi([w], [w_bf], [wt_bf], [ff], [fb])
    return w_bf, wt_bf, ff, fb


def test_fragment_major_images(dev):
    """Block (ct, ks) of an image = the 64-lane MFMA B fragment: lane l holds
    column ct*16 + (l & 15), reduce elements ks*32 + 8*(l >> 4) .. +7."""
    from nnx_ppo_amd import ops

    K, N = 70, 37
    w = torch.arange(K * N, dtype=torch.float32, device=dev).reshape(K, N) % 211 - 100
    w_bf, wt_bf, ff, fb = _shadows4(ops, w, dev)
    wb = w.to(BF)
    assert torch.equal(w_bf[:, :N], wb) and torch.equal(wt_bf[:, :K], wb.t())

    def defrag(img, C, R):
        NT, KS = (C + 15) // 16, (R + 31) // 32
        t = img.view(NT, KS, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(NT * 16, KS * 32)
        assert float(t[C:].float().abs().sum()) == 0 and float(t[:, R:].float().abs().sum()) == 0
        return t[:C, :R]

    assert torch.equal(defrag(ff, N, K), wb.t())  # forward: columns = outputs n, reduce = k
    assert torch.equal(defrag(fb, K, N), wb)      # backward: columns = k, reduce = n


def _make(obs, act, ah, ch, activation="relu", seed=17):
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    return factories.make_mlp_actor_critic(obs, act, ah, ch, Rngs(seed), activation=activation)


@pytest.mark.parametrize("activation", ["relu", "swish"])
def test_ppo_step_bf16_vs_oracle(dev, activation):
    from nnx_ppo_amd import config
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    N, T = 64, 12
    mk_env = lambda: EpisodeWrapper(MockEnv(5, 1, max_steps=5), 40)
    with config.use_compute_dtype("bf16"):
        env = mk_env()
        net = _make(5, 1, [64, 64, 64, 64], [256, 256], activation)
        ts = ppo.new_training_state(env, net, N, 18, 1e-3, device=dev)
        onet = on.from_product(net)
        oenv = mk_env()
        ots = op.new_training_state(oenv, onet, N, 18, keys, 1e-3)
        for k in range(2):
            ts, m = ppo.ppo_step(env, ts, N, T, 0.95, 0.99, 0.2, True, False, 2, 4)
            ots, info = op.ppo_step(oenv, ots, N, T, 0.95, 0.99, 0.2, True, 2, 4, keys)
            assert torch.equal(ts.env_states.obs.cpu(), ots.env_states.obs)  # events still exact
            a, c, r = (info[n].numpy().mean() for n in ("actor", "critic", "regularization"))
            assert np.allclose(m["losses/actor/mean"].item(), a, rtol=5e-2, atol=3e-4)
            assert np.allclose(m["losses/critic/mean"].item(), c, rtol=2e-3)
            assert np.allclose(m["losses/regularization/mean"].item(), r, rtol=5e-2, atol=1e-4)
        for p, q in zip(net.parameters(), onet.parameters()):
            assert float((p.data.cpu() - q.detach()).abs().max()) < 2e-2
            assert torch.isfinite(p.data).all()


def test_bf16_gradients_close_to_f32_path(dev):
    """Same minibatch through both MFMA paths: parameter gradients agree to bf16
    operand precision (cosine > 0.99 per tensor, relative L2 error < 0.15: the first actor layer sees bf16 obs)."""
    from nnx_ppo_amd import config
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.types import LoggingLevel, Transition
    from nnx_ppo_amd.networks.types import PPONetworkOutput
    from nnx_ppo_amd.optim import Optimizer

    T, B, O, A = 10, 256, 5, 1
    net = _make(O, A, [64, 64], [256, 256])
    net.to(dev)
    opt = Optimizer(net, 1e-4, device=dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=gen)
    done = torch.rand(T, B, device=dev, generator=gen) < 0.1
    extras = [rn(T, B, O), {"action": [None] * 3 + [rn(T, B, A)], "value": [None] * 3}]
    mb = Transition(obs=extras[0], network_output=PPONetworkOutput(None, rn(T, B) - 1, None),
                    rewards=rn(T, B), done=done, truncated=done & (rn(T, B) > 0),
                    next_obs=rn(1, B, O), metrics={}, rollout_extras=extras)
    grads = {}
    for mode in ("f32", "bf16"):
        with config.use_compute_dtype(mode):
            for s in [m for m in net.modules() if hasattr(m, "advance_rng")]:
                s._pending = 0  # same entropy noise for both passes
            opt.begin()
            ppo.ppo_loss(net, net.initialize_state(B), mb, 0.2, True, False, 0.99, 0.95, 1.0,
                         LoggingLevel.LOSSES)
            grads[mode] = [p.grad.clone() for p in net.parameters()]
    for (name, _), a, b in zip(net.named_parameters(), grads["f32"], grads["bf16"]):
        na = float(a.norm())
        if na == 0:
            continue
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        rel = float((a - b).norm() / a.norm())
        assert cos > 0.99 and rel < 0.15, (name, cos, rel)


@pytest.mark.parametrize("dims", [[5, 64, 64, 64, 64, 2], [5, 256, 256, 1], [17, 256, 256, 256, 256, 12],
                                  [17, 512, 512, 1], [33, 40, 129, 7], [3, 8]])
@pytest.mark.parametrize("M", [1, 1000, 4096, 9000])
@pytest.mark.parametrize("act", ["relu", "swish"])
def test_fused_mlp_forward_matches_per_layer_path(dev, dims, M, act):
    """The one-launch MLP trunk must reproduce the per-layer bf16 kernels (same
    MFMA, same k-order, same bf16 rounding points): outputs equal to fp32
    round-off, stored activations (y, pre-activation, bf16 input) identical."""
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(M + len(dims))
    L = len(dims) - 1
    code = ops.ACT_CODES[act]
    acts = [code] * (L - 1) + [ops.ACT_NONE]
    x = torch.as_tensor(rng.normal(size=(M, dims[0])).astype(np.float32)).to(dev)
    wts, ffs, biases = [], [], []
    for l in range(L):
        K, N = dims[l], dims[l + 1]
        w = torch.as_tensor((rng.normal(size=(K, N)) / math.sqrt(K)).astype(np.float32)).to(dev)
        _, wt_bf, ff, _ = _shadows4(ops, w, dev)
        wts.append(wt_bf)
        ffs.append(ff)
        biases.append(torch.as_tensor(rng.normal(size=N).astype(np.float32)).to(dev))
    out, saved = ops.mlp_fwd_bf16(x, ffs, biases, dims, acts, train=True)
    out_i, none = ops.mlp_fwd_bf16(x, ffs, biases, dims, acts, train=False)
    assert none is None and torch.allclose(out, out_i, rtol=1e-5, atol=1e-5)
    # per-layer reference path
    x_bf = ops.cast_pad_bf16(x)
    assert torch.equal(saved[0][0], x_bf)
    y = None
    for l in range(L):
        last = l == L - 1
        y, y_bf, pre = ops.dense_fwd_bf16(x_bf, wts[l], biases[l], dims[l], dims[l + 1], acts[l],
                                          want_f32=True, want_bf=True,
                                          want_preact=acts[l] == ops.ACT_SWISH)
        if not last:
            aux = pre if acts[l] == ops.ACT_SWISH else y_bf
            assert torch.equal(saved[l][1], aux), l
            assert torch.equal(saved[l + 1][0], y_bf), l
        x_bf = y_bf
    assert torch.allclose(out, y, rtol=1e-5, atol=1e-5)


def test_grouped_dw_matches_per_layer(dev):
    """All layers of a trunk in one grouped dW launch per tile class == the
    per-layer kernel, bit for bit (same tiles, same k-order, fixed-order slabs
    — only the number of splits differs, so compare to fp32 round-off)."""
    from nnx_ppo_amd import ops

    M = 3000
    rng = np.random.default_rng(0)
    shapes = [(5, 64), (64, 64), (64, 64), (64, 2), (5, 256), (256, 256), (256, 1), (17, 12)]
    problems, singles = [], []
    for K, N in shapes:
        x = ops.cast_pad_bf16(torch.as_tensor(rng.normal(size=(M, K)).astype(np.float32)).to(dev))
        dz = ops.cast_pad_bf16(torch.as_tensor(rng.normal(size=(M, N)).astype(np.float32)).to(dev))
        gw, gb = torch.zeros(K, N, device=dev), torch.zeros(N, device=dev)
        problems.append((x, dz, gw, gb if N != 12 else None))
        gw1, gb1 = torch.zeros(K, N, device=dev), torch.zeros(N, device=dev)
        ops.dense_bwd_dw_bf16(x, dz, gw1, gb1, accumulate=False)
        singles.append((gw1, gb1))
    ops.dense_bwd_dw_grouped_bf16(problems, accumulate=True)
    for (x, dz, gw, gb), (gw1, gb1), (K, N) in zip(problems, singles, shapes):
        want = x[:, :K].double().t() @ dz[:, :N].double()
        assert torch.allclose(gw.double(), want, rtol=1e-4, atol=2e-3), (K, N)
        assert torch.allclose(gw, gw1, rtol=1e-4, atol=2e-3), (K, N)
        if gb is not None:
            assert torch.allclose(gb, gb1, rtol=1e-4, atol=2e-3), (K, N)


@pytest.mark.parametrize("M", [40, 3000, 61440])
@pytest.mark.parametrize("clip", [None, 0.5])
def test_deferred_dw_requests_go_out_as_one_launch_and_reduce_inside_adam(dev, M, clip):
    """`Optimizer.begin(defer_dw=True)`: the dW requests of a gradient step are queued and go
    out as one grouped launch at `update()`, whose split-M slabs `mi_adam_step_slabs_f32`
    sums while it reads the gradient arena.  Against `begin()` (every request launched and
    reduced at once): same parameters, moments and bf16 images, gradient left zeroed."""
    from nnx_ppo_amd import ops, optim
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    results = []
    for defer in (False, True):
        net = factories.make_mlp_actor_critic(5, 2, [64, 64], [256], Rngs(3)).to(dev)
        opt = optim.Optimizer(net, 1e-3, gradient_clipping=clip, weight_decay=True)
        named = dict(net.named_parameters())
        layers = [(p, named.get(k[:-len("kernel")] + "bias")) for k, p in named.items()
                  if k.endswith("kernel") and len(p.shape) == 2]
        assert len(layers) == 5
        g = np.random.default_rng(7)
        opt.begin(defer_dw=defer)
        # something already in the gradient (the reduction ADDS to it)
        opt.grads.copy_(torch.as_tensor(g.normal(size=opt.n).astype(np.float32)).to(dev) * 1e-2)
        problems = []
        for w, bias in layers:
            K, N = w.shape
            x = ops.cast_pad_bf16(torch.as_tensor(g.normal(size=(M, K)).astype(np.float32)).to(dev))
            dz = ops.cast_pad_bf16(torch.as_tensor(g.normal(size=(M, N)).astype(np.float32)).to(dev) * 1e-2)
            problems.append((x, dz, w.grad, bias.grad if bias is not None else None))
        # two requests (as two modules would make them): queued, not launched, when deferring
        ops.dense_bwd_dw_grouped_bf16(problems[:2], accumulate=True)
        ops.dense_bwd_dw_grouped_bf16(problems[2:], accumulate=True)
        assert (len(ops.slab_defer.queue) == len(problems)) == defer
        assert ops.slab_defer.pending is None
        if clip is not None:
            opt.compute_grad_norm()  # flushes: a whole-gradient reader
            assert ops.slab_defer.pending is None and not ops.slab_defer.queue
            opt.update(have_norm=True)
        else:
            opt.update()
        assert ops.slab_defer.pending is None and ops.slab_defer.arena is None
        assert not ops.slab_defer.queue
        torch.cuda.synchronize()
        assert float(opt.grads.abs().max()) == 0.0
        from nnx_ppo_amd.networks import dense_chain
        results.append((opt.params.clone(), opt.m.clone(), opt.v.clone(),
                        [l._w_bf.clone() for l, _ in dense_chain.shadows_in(opt.params)]))
    a, b = results
    # (the non-deferring run makes two launches of 2 + 3 problems, the deferring one a single
    # launch of 5: the split-M plan differs, so equality is to fp32 summation order here;
    # bit-identity of the in-Adam reduction itself is the e2e test of test_train_loop_gpu)
    for x, y in zip(a[1:3], b[1:3]):  # moments: smooth in the gradient
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-6)
    # a first Adam step moves every parameter by ~ lr * sign(g): an element whose gradient
    # is within round-off of zero may land one step apart, nothing else may differ
    dp = (a[0] - b[0]).abs()
    assert float(dp.max()) <= 2.1e-3 and float((dp > 1e-6).float().mean()) < 1e-3
    for x, y in zip(a[3], b[3]):
        assert torch.allclose(x.float(), y.float(), rtol=1e-2, atol=1e-3)
    assert float((a[1] != 0).float().mean()) > 0.5


@pytest.mark.parametrize("dims", [[5, 64, 64, 64, 64, 2], [5, 256, 256, 1], [17, 512, 512, 1],
                                  [33, 40, 129, 7], [17, 256, 256, 256, 256, 12]])
@pytest.mark.parametrize("M", [7, 1000, 9000])
@pytest.mark.parametrize("act", ["relu", "tanh", "swish"])
@pytest.mark.parametrize("need_gin", [False, True])
def test_fused_mlp_backward_matches_per_layer_path(dev, dims, M, act, need_gin):
    """The one-launch dX chain must reproduce the per-layer dX kernels: every dz
    (bf16) identical up to one bf16 ulp of rounding-order noise, input gradient to
    fp32 round-off."""
    from nnx_ppo_amd import ops

    if act == "tanh" and M == 9000:
        pytest.skip("covered by relu / swish at this size")
    rng = np.random.default_rng(M + len(dims) + need_gin)
    L = len(dims) - 1
    code = ops.ACT_CODES[act]
    acts = [code] * (L - 1) + [ops.ACT_NONE]
    g = lambda a: torch.as_tensor(a.astype(np.float32)).to(dev)
    w_bfs, fbs, auxs = [], [], []
    for l in range(L):
        K, N = dims[l], dims[l + 1]
        w_bf, _, _, fb = _shadows4(ops, g(rng.normal(size=(K, N)) / math.sqrt(K)), dev)
        w_bfs.append(w_bf)
        fbs.append(fb)
        auxs.append(ops.cast_pad_bf16(g(rng.normal(size=(M, N)))))  # stand-in layer outputs
    g_out = g(rng.normal(size=(M, dims[-1])))
    dz, g_in = ops.mlp_bwd_dx_bf16(g_out, None, ops.ACT_NONE, fbs, dims, acts, auxs, need_gin)
    # per-layer reference
    ref = [None] * L
    ref[L - 1] = ops.cast_pad_bf16(g_out)
    assert torch.equal(dz[L - 1], ref[L - 1])
    for l in range(L - 1, 0, -1):
        _, ref[l - 1] = ops.dense_bwd_dx_bf16(ref[l], w_bfs[l], auxs[l - 1], acts[l - 1], dims[l],
                                              dims[l + 1], want_f32=False, want_bf=True)
        a, b = dz[l - 1].float(), ref[l - 1].float()
        assert torch.allclose(a, b, rtol=2 ** -7, atol=1e-6), l
        assert float(dz[l - 1][:, dims[l]:].float().abs().sum()) == 0
        assert (a != b).float().mean().item() < 0.02, l  # only rare 1-ulp differences
    if need_gin:
        want, _ = ops.dense_bwd_dx_bf16(ref[0], w_bfs[0], None, ops.ACT_NONE, dims[0], dims[1],
                                        want_f32=True, want_bf=False)
        assert torch.allclose(g_in, want, rtol=2e-2, atol=2e-2)
    else:
        assert g_in is None


def test_adam_keeps_the_bf16_shadows_fresh(dev):
    """mi_adam_step_f32 with shadows: after an update every bf16 image of a shadowed Dense
    kernel equals what mi_weights_to_bf16_multi derives from the new fp32 weights (bit
    for bit, padding included), so the forward pass needs no refresh launch."""
    from nnx_ppo_amd import config
    from nnx_ppo_amd.networks import dense_chain, factories
    from nnx_ppo_amd.networks.types import Rngs, param_epoch
    from nnx_ppo_amd.optim import Optimizer

    prev = config.compute_dtype()
    config.set_compute_dtype("bf16")
    try:
        net = factories.make_mlp([5, 64, 33, 7], Rngs(4), activation_last_layer=False)
        net.to(dev)
        opt = Optimizer(net, 1e-2, device=dev)
        layers = net.layers
        dense_chain.refresh(layers)                      # allocates and fills the shadows
        for step in range(3):
            opt.begin()
            opt.grads.normal_()
            opt.update()
            assert all(l._shadow_epoch == param_epoch() for l in layers)   # marked fresh
            got = [(l._w_bf.clone(), l._wt_bf.clone(), l._ff.clone(), l._fb.clone())
                   for l in layers]
            for l in layers:                             # recompute from the fp32 masters
                l._shadow_epoch = -1
            dense_chain.refresh(layers)
            for l, g in zip(layers, got):
                for a, b in zip(g, (l._w_bf, l._wt_bf, l._ff, l._fb)):
                    assert torch.equal(a, b)
    finally:
        config.set_compute_dtype(prev)
