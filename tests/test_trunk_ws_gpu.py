"""The weights-stationary trunk kernels (`csrc/trunk_ws.hip`) against the per-tile trunk
kernels of `csrc/mlp_bf16.hip`, which the other tests pin against the oracle: same MFMA
tiles, same accumulation order, same epilogue — every output and every kept bf16 image
must be identical bit for bit (`feedforward.py:42-51` chains, `containers.py:18-39`)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _trunk(dev, dims, seed):
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(seed)
    L = len(dims) - 1
    ffs, bs = [], []
    for l in range(L):
        K, N = dims[l], dims[l + 1]
        w = torch.tensor(rng.normal(size=(K, N)) / math.sqrt(K), dtype=torch.float32, device=dev)
        w_bf = torch.zeros(K, ops.pad8(N), dtype=BF, device=dev)
        wt_bf = torch.zeros(N, ops.pad8(K), dtype=BF, device=dev)
        nf, nb = ops.frag_sizes(K, N)
        ff, fb = torch.zeros(nf, dtype=BF, device=dev), torch.zeros(nb, dtype=BF, device=dev)
        ops.weights_to_bf16_multi([w], [w_bf], [wt_bf], [ff], [fb])
        ffs.append(ff)
        # biases live in a 256-byte aligned arena in the product; keep them aligned here too
        bs.append(torch.tensor(rng.normal(0, 0.3, size=N), dtype=torch.float32, device=dev))
    acts = [ops.ACT_RELU] * (L - 1) + [ops.ACT_NONE]
    return ffs, bs, acts


@pytest.mark.parametrize("dims", [[5, 256, 256, 1], [5, 64, 64, 64, 64, 2], [17, 128, 128, 3],
                                  [5, 64, 2], [32, 256, 16], [9, 128, 128, 128, 12],
                                  [1, 64, 64, 1]])
@pytest.mark.parametrize("M", [1, 63, 64, 1000, 16384, 30720, 31744])
def test_ws_forward_bit_identical_to_tile_kernels(dev, dims, M):
    from nnx_ppo_amd import ops

    ffs, bs, acts = _trunk(dev, dims, seed=M + len(dims))
    assert ops.mlp_ws_supported(dims, acts)
    x = torch.tensor(np.random.default_rng(M).normal(size=(M, dims[0])), dtype=torch.float32,
                     device=dev)
    want, saved0 = ops.mlp_fwd_bf16(x, ffs, bs, dims, acts, train=True)
    got, saved1 = ops.mlp_ws_fwd_bf16(x, ffs, bs, dims, acts, train=True)
    assert torch.equal(got, want), float((got - want).abs().max())
    for l, ((xa, ya), (xb, yb)) in enumerate(zip(saved0, saved1)):
        assert torch.equal(xa, xb), ("input image", l)
        if ya is not None or yb is not None:
            assert torch.equal(ya, yb), ("output image", l)
    inf, none = ops.mlp_ws_fwd_bf16(x, ffs, bs, dims, acts, train=False)
    assert none is None and torch.equal(inf, want)


def test_ws_shape_class(dev):
    from nnx_ppo_amd import ops

    R, N = ops.ACT_RELU, ops.ACT_NONE
    assert ops.mlp_ws_supported([5, 256, 256, 1], [R, R, N])
    assert ops.mlp_ws_supported([5, 64, 64, 64, 64, 2], [R, R, R, R, N])
    assert not ops.mlp_ws_supported([5, 256, 256, 256, 1], [R, R, R, N])   # registers
    assert not ops.mlp_ws_supported([5, 64, 128, 1], [R, R, N])            # unequal widths
    assert not ops.mlp_ws_supported([5, 64, 64, 1], [R, ops.ACT_TANH, N])  # relu only
    assert not ops.mlp_ws_supported([40, 64, 1], [R, N])                   # K0 <= 32
    assert not ops.mlp_ws_supported([5, 64, 17], [R, N])                   # N_out <= 16
    assert not ops.mlp_ws_supported([5, 512, 1], [R, N])


@pytest.mark.parametrize("shape", [(5, 1, [64] * 4, [256] * 2), (17, 6, [128] * 2, [128] * 3),
                                   (3, 2, [64], [64] * 2)])
@pytest.mark.parametrize("M,Mt", [(30720, 1024), (9000, 0), (12345, 77)])
def test_policy_ws_forward_bit_identical(dev, shape, M, Mt):
    """mi_policy_ws_fwd_bf16 (normaliser -> action trunk -> sampler; value trunk with the
    bootstrap tail rows) == mi_policy_fwd_bf16: log-likelihoods, regulariser rows, values,
    the sampler's input rows and every kept bf16 image, bit for bit."""
    from nnx_ppo_amd import ops

    O, A, ah, ch = shape
    a_dims, c_dims = [O] + ah + [2 * A], [O] + ch + [1]
    a_w, a_b, a_acts = _trunk(dev, a_dims, seed=M)
    c_w, c_b, c_acts = _trunk(dev, c_dims, seed=M + 1)
    assert ops.policy_ws_supported(a_dims, a_acts, c_dims, c_acts)
    rng = np.random.default_rng(M + Mt)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    obs = t(rng.normal(1.0, 2.0, size=(M, O)))
    tail = t(rng.normal(1.0, 2.0, size=(Mt, O))) if Mt else None
    extras = t(rng.normal(size=(M, A)))
    norm = (t(rng.normal(1.0, 0.5, size=O)), t(rng.uniform(50, 500, size=O)),
            torch.tensor(100.0, device=dev), 1e-6)
    rng_state = ops.make_rng_state(1234, dev, 7)
    kw = dict(min_std=0.1, std_scale=1.0, entropy_weight=1e-2, deterministic=False,
              extras=extras, train=True, want_stats=False, value_tail=tail)
    actor, critic = (a_w, a_b, a_dims, a_acts), (c_w, c_b, c_dims, c_acts)
    r0 = ops.policy_fwd_bf16(obs, norm, actor, critic, rng_state, 3, ws=False, **kw)
    r1 = ops.policy_fwd_bf16(obs, norm, actor, critic, rng_state, 3, ws=True, **kw)
    for k in ("log_likelihood", "reg", "value", "mean_and_std"):
        assert torch.equal(r0[k], r1[k]), (k, float((r0[k] - r1[k]).abs().max()))
    if Mt:
        assert torch.equal(r0["value_tail_out"], r1["value_tail_out"])
    for name in ("actor_saved", "critic_saved"):
        for l, ((xa, ya), (xb, yb)) in enumerate(zip(r0[name], r1[name])):
            assert torch.equal(xa, xb), (name, "x", l)
            assert (ya is None) == (yb is None) and (ya is None or torch.equal(ya, yb)), (name, l)
    # sampling (no stored actions): actions, raw draws and statistics too
    kw2 = dict(kw, extras=None, want_stats=True)
    s0 = ops.policy_fwd_bf16(obs, norm, actor, critic, rng_state, 5, ws=False, **kw2)
    s1 = ops.policy_fwd_bf16(obs, norm, actor, critic, rng_state, 5, ws=True, **kw2)
    for k in ("raw", "action", "log_likelihood", "reg", "mu", "sigma", "value"):
        assert torch.equal(s0[k], s1[k]), k


@pytest.mark.parametrize("shape", [(5, 1, [64] * 4, [256] * 2), (17, 6, [128] * 2, [128] * 2),
                                   (3, 2, [64] * 2, [64] * 2), (9, 3, [256] * 2, [256] * 2)])
@pytest.mark.parametrize("M,Mt", [(4096, 0), (1, 0), (33, 5), (8000, 192), (2048, 0)])
@pytest.mark.parametrize("train", [False, True])
def test_policy_ws_one_launch_at_rollout_sizes(dev, shape, M, Mt, train):
    """Up to 8192 rows mi_policy_ws_fwd_bf16 runs both trunks in ONE launch
    (policy_ws_dual_kernel): sampled actions, raw draws, log-likelihoods, statistics and
    values == mi_policy_fwd_bf16, bit for bit — sampling and replay, with and without the
    kept images."""
    from nnx_ppo_amd import _lib, ops

    O, A, ah, ch = shape
    a_dims, c_dims = [O] + ah + [2 * A], [O] + ch + [1]
    a_w, a_b, a_acts = _trunk(dev, a_dims, seed=M)
    c_w, c_b, c_acts = _trunk(dev, c_dims, seed=M + 1)
    assert ops.policy_ws_dual_supported(a_dims, a_acts, c_dims, c_acts)
    rng = np.random.default_rng(M + Mt)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    obs = t(rng.normal(1.0, 2.0, size=(M, O)))
    tail = t(rng.normal(1.0, 2.0, size=(Mt, O))) if Mt else None
    norm = (t(rng.normal(1.0, 0.5, size=O)), t(rng.uniform(50, 500, size=O)),
            torch.tensor(100.0, device=dev), 1e-6)
    rng_state = ops.make_rng_state(99, dev, 11)
    actor, critic = (a_w, a_b, a_dims, a_acts), (c_w, c_b, c_dims, c_acts)
    for extras in (None, t(rng.normal(size=(M, A)))):
        for det in (False, True):
            kw = dict(min_std=0.1, std_scale=1.0, entropy_weight=1e-2, deterministic=det,
                      extras=extras, train=train, want_stats=True, value_tail=tail)
            r0 = ops.policy_fwd_bf16(obs, norm, actor, critic, rng_state, 5, ws=False, **kw)
            with _lib.profiler as prof:
                r1 = ops.policy_fwd_bf16(obs, norm, actor, critic, rng_state, 5, ws=True, **kw)
            assert [r[0] for r in prof.records] == ["mi_policy_ws_fwd_bf16"]
            for k in ("raw", "action", "log_likelihood", "reg", "mu", "sigma", "value"):
                if r0[k] is None:
                    assert r1[k] is None or k == "action"
                    continue
                assert torch.equal(r0[k], r1[k]), (k, det, extras is None)
            if Mt:
                assert torch.equal(r0["value_tail_out"], r1["value_tail_out"])
            if train:
                assert torch.equal(r0["mean_and_std"], r1["mean_and_std"])
                for name in ("actor_saved", "critic_saved"):
                    for (xa, ya), (xb, yb) in zip(r0[name], r1[name]):
                        assert torch.equal(xa, xb)
                        assert (ya is None) == (yb is None) and (ya is None or torch.equal(ya, yb))


def test_policy_ws_dual_menu(dev):
    from nnx_ppo_amd import ops

    R, N = ops.ACT_RELU, ops.ACT_NONE
    ok = lambda a, c: ops.policy_ws_dual_supported(a, [R] * (len(a) - 2) + [N],
                                                   c, [R] * (len(c) - 2) + [N])
    assert ok([5, 64, 64, 64, 64, 2], [5, 256, 256, 1])       # BASELINE C2
    assert ok([5, 128, 128, 4], [5, 128, 128, 1])
    assert not ok([5, 64, 64, 64, 64, 2], [5, 128, 128, 1])   # not instantiated: tile kernel
    assert not ok([5, 512, 2], [5, 256, 256, 1])              # outside the shape class


def test_ppo_step_on_ws_kernels_equals_tile_kernels(dev):
    """A whole iteration at a training size with the loss replay on the weights-stationary
    kernels == the same iteration on the per-tile kernels."""
    from nnx_ppo_amd import config
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import cartpole_shaped
    from nnx_ppo_amd.networks import factories, policy
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    outs = []
    with config.use_compute_dtype("bf16"):
        for ws in (True, False):
            policy.WS_POLICY = policy.WS_POLICY_BWD = ws
            try:
                env = EpisodeWrapper(cartpole_shaped(max_steps=9), 30)
                net = factories.make_mlp_actor_critic(5, 1, [64] * 4, [256] * 2, Rngs(17))
                ts = ppo.new_training_state(env, net, 2048, 17, 1e-3, device=dev)
                for _ in range(2):  # M = 10 * 1024 + 1024 rows per gradient step
                    ts, m = ppo.ppo_step(env, ts, 2048, 10, 0.95, 0.99, 0.2, True, False, 2, 2)
                outs.append((ts.optimizer.params.clone(), {k: float(v) for k, v in m.items()}))
            finally:
                policy.WS_POLICY, policy.WS_POLICY_BWD = True, True
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]


def _bwd_images(dev, dims, seed):
    """Backward fragment-major images of a random trunk (+ the forward ones, biases)."""
    from nnx_ppo_amd import ops

    rng = np.random.default_rng(seed)
    L = len(dims) - 1
    ffs, fbs, bs = [], [], []
    for l in range(L):
        K, N = dims[l], dims[l + 1]
        w = torch.tensor(rng.normal(size=(K, N)) / math.sqrt(K), dtype=torch.float32, device=dev)
        w_bf = torch.zeros(K, ops.pad8(N), dtype=BF, device=dev)
        wt_bf = torch.zeros(N, ops.pad8(K), dtype=BF, device=dev)
        nf, nb = ops.frag_sizes(K, N)
        ff, fb = torch.zeros(nf, dtype=BF, device=dev), torch.zeros(nb, dtype=BF, device=dev)
        ops.weights_to_bf16_multi([w], [w_bf], [wt_bf], [ff], [fb])
        ffs.append(ff)
        fbs.append(fb)
        bs.append(torch.tensor(rng.normal(0, 0.3, size=N), dtype=torch.float32, device=dev))
    acts = [ops.ACT_RELU] * (L - 1) + [ops.ACT_NONE]
    return ffs, fbs, bs, acts


@pytest.mark.parametrize("dims", [[5, 256, 256, 1], [5, 64, 64, 64, 64, 2], [17, 128, 128, 3],
                                  [5, 64, 2], [9, 128, 128, 128, 12]])
@pytest.mark.parametrize("M", [1, 63, 1000, 16384, 30720])
def test_ws_backward_bit_identical_to_tile_kernels(dev, dims, M):
    """mi_mlp_ws_bwd_dx_bf16 == mi_mlp_bwd_dx_bf16 (no input gradient): every dz image."""
    from nnx_ppo_amd import ops

    ffs, fbs, bs, acts = _bwd_images(dev, dims, seed=M + len(dims))
    rng = np.random.default_rng(M)
    x = torch.tensor(rng.normal(size=(M, dims[0])), dtype=torch.float32, device=dev)
    _, saved = ops.mlp_fwd_bf16(x, ffs, bs, dims, acts, train=True)
    auxs = [sv[1] for sv in saved]
    g = torch.tensor(rng.normal(size=(M, dims[-1])), dtype=torch.float32, device=dev)
    want, _ = ops.mlp_bwd_dx_bf16(g, None, ops.ACT_NONE, fbs, dims, acts, auxs, False)
    got = ops.mlp_ws_bwd_dx_bf16(g, fbs, dims, acts, auxs)
    for l, (a, b) in enumerate(zip(want, got)):
        assert torch.equal(a, b), (l, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize("shape", [(5, 1, [64] * 4, [256] * 2), (17, 6, [128] * 2, [128] * 3)])
@pytest.mark.parametrize("M", [30720, 9000, 12345])
def test_policy_ws_backward_bit_identical(dev, shape, M):
    """mi_policy_ws_bwd_bf16 == mi_policy_bwd_bf16: the sampler backward's rows and every
    dz of both trunks."""
    from nnx_ppo_amd import ops

    O, A, ah, ch = shape
    a_dims, c_dims = [O] + ah + [2 * A], [O] + ch + [1]
    a_ff, a_fb, a_b, a_acts = _bwd_images(dev, a_dims, seed=M)
    c_ff, c_fb, c_b, c_acts = _bwd_images(dev, c_dims, seed=M + 1)
    rng = np.random.default_rng(M)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    obs, extras = t(rng.normal(size=(M, O))), t(rng.normal(size=(M, A)))
    rng_state = ops.make_rng_state(99, dev, 3)
    kw = dict(min_std=0.1, std_scale=1.0, entropy_weight=1e-2)
    r = ops.policy_fwd_bf16(obs, None, (a_ff, a_b, a_dims, a_acts), (c_ff, c_b, c_dims, c_acts),
                            rng_state, 2, deterministic=False, extras=extras, train=True,
                            want_stats=False, **kw)
    g_ll, g_v = t(rng.normal(size=M) / M), t(rng.normal(size=(M, 1)) / M)
    actor = (a_fb, a_dims, a_acts, [sv[1] for sv in r["actor_saved"]])
    critic = (c_fb, c_dims, c_acts, [sv[1] for sv in r["critic_saved"]])
    want = ops.policy_bwd_bf16(r["mean_and_std"], extras, rng_state, 2, g_ll, 1.0 / M, g_v,
                               actor, critic, ws=False, **kw)
    got = ops.policy_bwd_bf16(r["mean_and_std"], extras, rng_state, 2, g_ll, 1.0 / M, g_v,
                              actor, critic, ws=True, **kw)
    for name, wl, gl in (("actor", want[0], got[0]), ("critic", want[1], got[1])):
        for l, (a, b) in enumerate(zip(wl, gl)):
            assert torch.equal(a, b), (name, l, float((a.float() - b.float()).abs().max()))
    # the same through the relu' masks of the weights-stationary forward
    rw = ops.policy_fwd_bf16(obs, None, (a_ff, a_b, a_dims, a_acts), (c_ff, c_b, c_dims, c_acts),
                             rng_state, 2, deterministic=False, extras=extras, train=True,
                             want_stats=False, ws=True, **kw)
    masks = (rw["actor_masks"], rw["critic_masks"])
    assert all(m is not None for m in masks[0][:-1] + masks[1][:-1])
    # format: tile (rt, ct) has byte [rt >> 2][ct][lane][rt & 3];
    # bit e <-> y[16 rt + (lane & 15)][16 ct + 4 (lane >> 4) + e] > 0
    for saved, ms in ((rw["actor_saved"], masks[0]), (rw["critic_saved"], masks[1])):
        for l, m in enumerate(ms[:-1]):
            n_ct = m.shape[1]
            y = saved[l][1][:, :n_ct * 16].float()                 # [M, N]: aux of layer l
            rows = (M // 64) * 64
            if rows == 0:
                continue
            pos = (y[:rows] > 0).view(rows // 64, 4, 16, n_ct, 4, 4)  # g, rt&3, li, ct, lq, e
            want_bits = pos.permute(0, 3, 4, 2, 1, 5).reshape(rows // 64, n_ct, 64, 4, 4)
            bits = torch.arange(4, device=dev)
            got_bits = ((m[:rows // 64].unsqueeze(-1).to(torch.int32) >> bits) & 1).bool()
            assert torch.equal(got_bits, want_bits), l
    gotm = ops.policy_bwd_bf16(rw["mean_and_std"], extras, rng_state, 2, g_ll, 1.0 / M, g_v,
                               actor, critic, ws=True, masks=masks, **kw)
    for name, wl, gl in (("actor", want[0], gotm[0]), ("critic", want[1], gotm[1])):
        for l, (a, b) in enumerate(zip(wl, gl)):
            assert torch.equal(a, b), ("masks", name, l)


@pytest.mark.parametrize("shape", [(5, 1, [64] * 4, [256] * 2), (17, 6, [128] * 2, [128] * 2),
                                   (3, 2, [64] * 3, [64] * 3)])
@pytest.mark.parametrize("T,B", [(30, 1024), (30, 512), (7, 2048), (32, 1984)])
@pytest.mark.parametrize("normalize,with_reg", [(True, True), (False, False)])
def test_policy_ws_backward_with_gae_inside(dev, shape, T, B, normalize, with_reg):
    """mi_policy_ws_bwd_gae_bf16 == mi_gae_ppo_loss_f32 followed by mi_policy_ws_bwd_bf16
    (ppo.py:433-503 + the backward): every dz image of both trunks bit for bit, the four
    loss scalars to fp64 summation order; twice, so the re-armed ticket is exercised."""
    if not _gae_inside_case(dev, shape, T, B, normalize, with_reg):
        pytest.skip("outside the fused class on this chip")


def test_policy_ws_backward_with_gae_inside_many_tiles_per_workgroup():
    """The same with the action trunk on 38 % of the CUs (MIPPO_WS_GAE_SPLIT, read once per
    process: a child process): 1024 row tiles over 97 action-trunk workgroups = 11 each, so
    the sampler backward's second stash round (tiles 9 .. 11 of a workgroup) is exercised."""
    import os
    import subprocess
    import sys

    code = ("import sys, torch; sys.path.insert(0, 'tests'); import test_trunk_ws_gpu as t; "
            "ok = t._gae_inside_case(torch.device('cuda:0'), (5, 1, [64] * 4, [256] * 2), 32, "
            "2048, True, True); print('CASE', ok)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True,
                       env=dict(os.environ, MIPPO_WS_GAE_SPLIT="62"), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "CASE True" in r.stdout, r.stdout + r.stderr


def _gae_inside_case(dev, shape, T, B, normalize, with_reg) -> bool:
    from nnx_ppo_amd import ops

    O, A, ah, ch = shape
    M = T * B
    a_dims, c_dims = [O] + ah + [2 * A], [O] + ch + [1]
    a_ff, a_fb, a_b, a_acts = _bwd_images(dev, a_dims, seed=M)
    c_ff, c_fb, c_b, c_acts = _bwd_images(dev, c_dims, seed=M + 1)
    rng = np.random.default_rng(M + T)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    obs, extras = t(rng.normal(size=(M, O))), t(rng.normal(size=(M, A)))
    tail = t(rng.normal(size=(B, O)))
    rng_state = ops.make_rng_state(99, dev, 3)
    kw = dict(min_std=0.1, std_scale=1.0, entropy_weight=1e-2)
    rw = ops.policy_fwd_bf16(obs, None, (a_ff, a_b, a_dims, a_acts), (c_ff, c_b, c_dims, c_acts),
                             rng_state, 2, deterministic=False, extras=extras, train=True,
                             want_stats=False, ws=True, value_tail=tail, **kw)
    masks = (rw["actor_masks"], rw["critic_masks"])
    actor = (a_fb, a_dims, a_acts, [sv[1] for sv in rw["actor_saved"]])
    critic = (c_fb, c_dims, c_acts, [sv[1] for sv in rw["critic_saved"]])
    if not ops.policy_bwd_gae_supported(T, B, actor, critic):
        return False
    values = rw["value"].view(T, B).contiguous()
    last_value = rw["value_tail_out"].view(B).contiguous()
    ll_new = rw["log_likelihood"].view(T, B).contiguous()
    ll_old = (ll_new + t(rng.normal(0, 0.3, size=(T, B)))).contiguous()
    reg = rw["reg"].view(T, B).contiguous() if with_reg else None
    rewards = t(rng.normal(size=(T, B)))
    done = torch.tensor(rng.random((T, B)) < 0.15, device=dev)
    trunc = torch.tensor(rng.random((T, B)) < 0.05, device=dev) & done
    args = (0.99, 0.95, normalize, 0.2, 0.5)
    g_ll, g_v, want_loss, _ = ops.gae_ppo_loss(rewards, values, last_value, done, trunc, ll_new,
                                               ll_old, reg, *args)
    want = ops.policy_bwd_bf16(rw["mean_and_std"], extras, rng_state, 2, g_ll.view(M), 1.0 / M,
                               g_v.view(M, 1), actor, critic, ws=True, masks=masks, **kw)
    for rep in range(2):
        got_a, got_c, got_loss = ops.policy_bwd_gae_bf16(
            rw["mean_and_std"], extras, rng_state, 2, 1.0 / M, actor, critic, masks, rewards,
            values, last_value, done, trunc, ll_new, ll_old, reg, *args, **kw)
        for name, wl, gl in (("actor", want[0], got_a), ("critic", want[1], got_c)):
            for l, (a, b) in enumerate(zip(wl, gl)):
                assert torch.equal(a, b), (rep, name, l,
                                           float((a.float() - b.float()).abs().max()))
        assert torch.allclose(got_loss, want_loss, rtol=1e-6, atol=1e-9), (got_loss, want_loss)
    # deferred: the launch leaves its per-tile partials, mi_policy_loss_finalize_f32 sums them
    # (two pending launches in one finalize): the same bits as the sum at the launch's tail
    pending: list = []
    outs = []
    for rep in range(2):
        lo = torch.full((4,), float("nan"), device=dev)
        d_a, d_c, lo2 = ops.policy_bwd_gae_bf16(
            rw["mean_and_std"], extras, rng_state, 2, 1.0 / M, actor, critic, masks, rewards,
            values, last_value, done, trunc, ll_new, ll_old, reg, *args, loss_out=lo,
            defer=pending, **kw)
        assert lo2 is lo
        outs.append((d_a, d_c, lo))
    assert len(pending) == 2
    ops.policy_loss_finalize(pending)
    assert not pending
    for d_a, d_c, lo in outs:
        assert torch.equal(lo, got_loss), (lo, got_loss)
        for wl, gl in ((got_a, d_a), (got_c, d_c)):
            for a, b in zip(wl, gl):
                assert torch.equal(a, b)
    return True


def test_ppo_step_with_gae_in_backward_equals_separate_launch(dev):
    """A whole iteration with the GAE / loss inside the backward launch == the same iteration
    with the mi_gae_ppo_loss_f32 launch: parameters bit for bit, logged losses to 1e-6."""
    from nnx_ppo_amd import config
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import cartpole_shaped
    from nnx_ppo_amd.networks import factories, policy
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    outs = []
    with config.use_compute_dtype("bf16"):
        for fused in (True, False):
            policy.GAE_IN_BWD = fused
            try:
                env = EpisodeWrapper(cartpole_shaped(max_steps=9), 30)
                net = factories.make_mlp_actor_critic(5, 1, [64] * 4, [256] * 2, Rngs(17))
                ts = ppo.new_training_state(env, net, 2048, 17, 1e-3, device=dev)
                for _ in range(2):  # [T, B] = [30, 1024] per gradient step
                    ts, m = ppo.ppo_step(env, ts, 2048, 30, 0.95, 0.99, 0.2, True, False, 2, 2)
                outs.append((ts.optimizer.params.clone(), {k: float(v) for k, v in m.items()}))
            finally:
                policy.GAE_IN_BWD = True
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1].keys() == outs[1][1].keys()
    for k, v in outs[0][1].items():
        assert v == pytest.approx(outs[1][1][k], rel=1e-6, abs=1e-9), k
    from nnx_ppo_amd import ops

    assert ops.policy_bwd_gae_timeouts() == 0  # no statistics hand-over ever timed out
