"""Integer key kernels and episode bookkeeping: the HIP kernels must reproduce
the integer torch expressions of nnx_ppo_amd/random.py (evaluated on CPU) bit
for bit — this is what makes env resets, minibatch permutations and episode
counters identical between the CPU oracle run and the GPU run."""
import pytest
import torch

from nnx_ppo_amd import random as keys

pytestmark = pytest.mark.gpu


def test_key_ops_bit_exact_vs_cpu(dev):
    k = keys.key(1234)
    kc = keys.split(k, 1000)            # CPU path
    kg = keys.split(k.to(dev), 1000)    # HIP path
    assert torch.equal(kg.cpu(), kc)
    assert torch.equal(keys.split(kg, (3, 2)).cpu(), keys.split(kc, (3, 2)))
    assert torch.equal(keys.bits(kg, (7,)).cpu(), keys.bits(kc, (7,)))
    assert torch.equal(keys.randint(kg, (), 0, 500).cpu(), keys.randint(kc, (), 0, 500))
    assert torch.equal(keys.randint(kg, (4,), -3, 11).cpu(), keys.randint(kc, (4,), -3, 11))
    assert torch.equal(keys.uniform(kg, (5,)).cpu(), keys.uniform(kc, (5,)))
    assert torch.equal(keys.unit_uniform(kg, (17,)).cpu(), keys.unit_uniform(kc, (17,)))
    data = torch.arange(1000, dtype=torch.int64) * 7 - 3
    assert torch.equal(keys.fold_key(kg, data.to(dev)).cpu(), keys.fold_key(kc, data))
    assert int(keys.fold_in(k.to(dev), 5)) == int(keys.fold_in(k, 5))
    p_g = keys.permutation(keys.fold_in(k.to(dev), 2), 4096)
    assert torch.equal(p_g.cpu(), keys.permutation(keys.fold_in(k, 2), 4096))
    # scalar key and empty shapes
    assert torch.equal(keys.split(k.to(dev)).cpu(), keys.split(k))
    assert keys.split(kg[:0], 3).shape == (0, 3)


def test_envs_identical_on_cpu_and_gpu(dev):
    from nnx_ppo_amd.envs import DummyCounterEnv, MockEnv, cheetah_shaped
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    for mk in (lambda: EpisodeWrapper(MockEnv(5, 1, max_steps=4), 9),
               lambda: EpisodeWrapper(cheetah_shaped(max_steps=6), 7),
               lambda: DummyCounterEnv()):
        ec, eg = mk(), mk()
        kc = keys.split(keys.key(3), 64)
        sc, sg = ec.reset(kc), eg.reset(kc.to(dev))
        for _ in range(12):
            a = torch.zeros(64, 1)
            sc, sg = ec.step(sc, a), eg.step(sg, a.to(dev))
            lc = [x for x in _leaves(sc)]
            lg = [x for x in _leaves(sg)]
            assert len(lc) == len(lg)
            for x, y in zip(lc, lg):
                assert x.dtype == y.dtype and torch.equal(x, y.cpu())


def _leaves(s):
    from nnx_ppo_amd.tree import tree_leaves

    return tree_leaves(s)


def test_episode_step_kernel(dev):
    from nnx_ppo_amd import ops

    n = 1000
    g = torch.Generator().manual_seed(0)
    c = torch.randint(0, 12, (n,), generator=g)
    d_f = (torch.rand(n, generator=g) < 0.2).float()
    tr = torch.rand(n, generator=g) < 0.1
    for done, trunc in ((d_f, None), (d_f != 0, tr), (d_f, tr)):
        co, to, do, fo = ops.episode_step(c.to(dev), done.to(dev),
                                          None if trunc is None else trunc.to(dev), 10)
        wc = c + 1
        wt = (wc >= 10) | (trunc if trunc is not None else torch.zeros(n, dtype=torch.bool))
        wd = ((done != 0) | wt).float()
        assert torch.equal(co.cpu(), wc) and torch.equal(to.cpu(), wt) and torch.equal(do.cpu(), wd)
        assert to.dtype == torch.bool and do.dtype == torch.float32
        assert fo.dtype == torch.bool and torch.equal(fo.cpu(), wd != 0)


def test_unit_uniform_fold_matches_two_step(dev):
    """key_expand's in-launch fold == fold_key then expand, on GPU and on CPU."""
    from nnx_ppo_amd import random as rnd

    k = rnd.split(rnd.key(5), 300)
    step = torch.arange(300, dtype=torch.int64) * 7 - 11
    want = rnd.unit_uniform(rnd.fold_key(k, step), (6,))
    assert torch.equal(rnd.unit_uniform(k, (6,), fold=step), want)
    got = rnd.unit_uniform(k.to(dev), (6,), fold=step.to(dev))
    assert torch.equal(got.cpu(), want)


def test_select_rows_multi(dev):
    from nnx_ppo_amd import ops

    B = 257
    g = torch.Generator().manual_seed(2)
    mask = torch.rand(B, generator=g) < 0.4
    pairs = []
    for i in range(19):  # > 16: exercises the batching
        shape = [(B,), (B, 5), (B, 3, 2), (B, 64)][i % 4]
        dt = [torch.float32, torch.int64, torch.bool, torch.uint8][i % 4]
        mk = lambda: (torch.randn(shape, generator=g) * 50).to(dt)
        a, b = mk(), mk()
        if i % 5 == 4 and len(shape) > 1:
            a = a[0].clone()  # broadcast row
        pairs.append((a, b))
    outs = ops.select_rows_multi(mask.to(dev), [(a.to(dev), b.to(dev)) for a, b in pairs])
    for (a, b), o in zip(pairs, outs):
        m = mask.reshape(B, *([1] * (b.dim() - 1)))
        assert torch.equal(o.cpu(), torch.where(m, a if a.shape == b.shape else a.expand_as(b), b))


def test_split2_and_gather_multi(dev):
    from nnx_ppo_amd import ops

    k = keys.split(keys.key(5), 300)
    a_c, b_c = keys.split2(k)
    a_g, b_g = keys.split2(k.to(dev))
    assert a_g.is_contiguous() and b_g.is_contiguous()
    assert torch.equal(a_g.cpu(), a_c) and torch.equal(b_g.cpu(), b_c)
    assert torch.equal(a_c, keys.split(k)[..., 0]) and torch.equal(b_c, keys.split(k)[..., 1])
    g = torch.Generator().manual_seed(0)
    N, L = 200, 48
    leaves = []
    for i in range(19):
        T = [7, 1, 3][i % 3]
        feat = [(), (5,), (2, 3), (64,)][i % 4]
        dt = [torch.float32, torch.bool, torch.int64, torch.uint8][i % 4]
        leaves.append((torch.randn(T, N, *feat, generator=g) * 40).to(dt))
    idx = torch.randperm(N, generator=g)[:L]
    outs = ops.gather_cols_multi([x.to(dev) for x in leaves], idx.to(dev))
    for x, o in zip(leaves, outs):
        assert torch.equal(o.cpu(), x[:, idx])


def test_copy_multi_bit_exact(dev):
    """mi_copy_multi: several buffers (odd sizes, every dtype) in one launch."""
    from nnx_ppo_amd import ops

    g = torch.Generator().manual_seed(4)
    srcs = []
    for i in range(19):  # > 16: exercises the batching
        shape = [(), (7,), (4096,), (4096, 5), (3, 3, 3)][i % 5]
        dt = [torch.float32, torch.int64, torch.bool, torch.uint8][i % 4]
        srcs.append((torch.randn(shape, generator=g) * 50).to(dt).to(dev))
    dsts = [torch.zeros_like(s) for s in srcs]
    ops.copy_multi(list(zip(dsts, srcs)))
    for d, s in zip(dsts, srcs):
        assert torch.equal(d, s)


@pytest.mark.parametrize("n", [1, 2, 5, 100, 1000, 4096, 5000, 8192])
def test_key_permutations_match_the_host_scheme(dev, n):
    """mi_key_permutations == stack(permutation(fold_in(key, e), n)) of the integer
    torch ops (stable argsort of the hashes), bit for bit; rows are permutations."""
    from nnx_ppo_amd import random as rnd

    k = rnd.key(12345 + n)
    want = rnd.permutations(k, 4, n)                       # CPU: fold_in + bits + argsort
    assert want.shape == (4, n)
    got = rnd.permutations(k.to(dev), 4, n)
    assert got.dtype == torch.int64 and torch.equal(got.cpu(), want)
    for e in range(4):
        assert torch.equal(want[e], rnd.permutation(rnd.fold_in(k, e), n))
        assert torch.equal(torch.sort(got[e].cpu()).values, torch.arange(n))


@pytest.mark.parametrize("B", [1, 257, 4096])
def test_episode_step_select_equals_step_then_select(dev, B):
    """`mi_episode_step_select` == `mi_episode_step` followed by `mi_select_rows_multi` on
    the done flag, bit for bit: the wrapper's own leaves and generic leaves of 1-, 4- and
    20-byte rows (episode_wrapper.py:12-22, rollout.py:41-44)."""
    from nnx_ppo_amd import ops

    g = torch.Generator().manual_seed(B)
    counter = torch.randint(0, 12, (B,), generator=g).to(dev)
    inner_done = (torch.rand(B, generator=g) < 0.2).to(dev)
    inner_trunc = (torch.rand(B, generator=g) < 0.1).to(dev)
    rc = torch.randint(0, 5, (B,), generator=g).to(dev)
    rt = torch.zeros(B, dtype=torch.bool, device=dev)
    rd = torch.zeros(B, dtype=torch.float32, device=dev)
    leaves = [(torch.randn(B, 5, generator=g).to(dev), torch.randn(B, 5, generator=g).to(dev)),
              (torch.randn(B, generator=g).to(dev), torch.randn(B, generator=g).to(dev)),
              ((torch.rand(B, generator=g) < 0.5).to(dev), (torch.rand(B, generator=g) < 0.5).to(dev)),
              (torch.randint(0, 99, (B,), generator=g).to(dev),
               torch.randint(0, 99, (B,), generator=g).to(dev))]
    for trunc in (inner_trunc, None):
        for done_in in (inner_done, inner_done.float()):
            c0, t0, d0, f0 = ops.episode_step(counter, done_in, trunc, 10)
            want = ops.select_rows_multi(f0, leaves + [(rc, c0), (rt, t0), (rd, d0)])
            c, t, d, f, cs, ts, ds, outs = ops.episode_step_select(counter, done_in, trunc, 10,
                                                                   rc, rt, rd, leaves)
            for a, b in zip((c, t, d, f), (c0, t0, d0, f0)):
                assert a.dtype == b.dtype and torch.equal(a, b)
            for a, b in zip(outs + [cs, ts, ds], want):
                assert a.dtype == b.dtype and torch.equal(a, b)


def test_stack_multi_equals_torch_stack(dev):
    """`mi_stack_multi` == `torch.stack(group, 0)` per group, bit for bit: 4-byte and 1-byte
    rows, a shared (repeated) source, more groups than one launch takes."""
    from nnx_ppo_amd import ops

    g = torch.Generator().manual_seed(5)
    T = 30
    shared = torch.randn(64, generator=g).to(dev)
    groups = [[torch.randn(64, 5, generator=g).to(dev) for _ in range(T)],
              [(torch.rand(64, generator=g) < 0.5).to(dev) for _ in range(T)],
              [torch.randint(0, 9, (64,), generator=g).to(dev) for _ in range(T)],
              [shared] * T,
              [torch.randn(3, generator=g).to(dev) for _ in range(T)]] * 4
    got = ops.stack_multi(groups)
    for a, grp in zip(got, groups):
        want = torch.stack(grp, 0)
        assert a.dtype == want.dtype and a.shape == want.shape and torch.equal(a, want)
    assert ops.stack_multi([]) == []


def test_gather_cols_multi_groups(dev):
    """`groups` > 1: group g of every output == the single-group gather of that group's
    indices (each minibatch a contiguous time-major block, ppo.py:284-300)."""
    from nnx_ppo_amd import ops

    g = torch.Generator().manual_seed(11)
    T, N, G, mb = 7, 96, 6, 32
    leaves = [torch.randn(T, N, 5, generator=g).to(dev), torch.randn(T, N, generator=g).to(dev),
              (torch.rand(T, N, generator=g) < 0.5).to(dev),
              torch.randint(0, 9, (1, N), generator=g).to(dev)]
    idx = torch.stack([torch.randperm(N, generator=g)[:mb] for _ in range(G)]).to(dev)
    got = ops.gather_cols_multi(leaves, idx.reshape(-1).contiguous(), groups=G)
    for k in range(G):
        want = ops.gather_cols_multi(leaves, idx[k].contiguous())
        for a, b, src in zip(got, want, leaves):
            assert a.shape == (G, src.shape[0], mb, *src.shape[2:]) and a[k].is_contiguous()
            assert torch.equal(a[k], b) and torch.equal(b, src[:, idx[k]])


@pytest.mark.parametrize("obs_size", [5, 1, {"position": 8, "velocity": 9}, {"a": 1, "b": 2, "c": 3}])
def test_mock_env_step_one_launch_equals_host_statement(dev, obs_size):
    """MockEnv.step on the GPU is ONE launch (mi_mock_env_step); the CPU path is the plain
    torch statement of the same env (`test_dummies/mock_env.py:25-63` restated): counters,
    done flags and every observation leaf agree bit for bit over an episode."""
    from nnx_ppo_amd import _lib
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.tree import tree_leaves

    n = 1000
    k_cpu = keys.split(keys.key(9), n)
    k_gpu = k_cpu.to(dev)
    env = MockEnv(obs_size, 2, max_steps=4)
    s_cpu, s_gpu = env.reset(k_cpu), env.reset(k_gpu)
    for _ in range(6):
        with _lib.profiler as prof:
            s_gpu = env.step(s_gpu, None)
        assert [r[0] for r in prof.records] == ["mi_mock_env_step"]
        s_cpu = env.step(s_cpu, None)
        for a, b in zip(tree_leaves(s_gpu), tree_leaves(s_cpu)):
            assert a.dtype == b.dtype and torch.equal(a.cpu(), b), (a.dtype, b.dtype)
    assert bool(s_gpu.done.all())


@pytest.mark.parametrize("obs_size", [5, {"position": 8, "velocity": 9}])
def test_mock_env_step_inside_the_episode_launch(dev, obs_size):
    """EpisodeWrapper(MockEnv).step_and_reset runs the env's own step inside the wrapper's
    launch (mi_mock_episode_step_select): stepped state and state after the reset select
    equal the two-launch path (mi_mock_env_step, then mi_episode_step_select) bit for bit,
    across inner-env dones and wrapper truncations."""
    from nnx_ppo_amd import _lib
    from nnx_ppo_amd.envs import MockEnv
    from nnx_ppo_amd.tree import tree_leaves
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    n = 777
    fused = EpisodeWrapper(MockEnv(obs_size, 2, max_steps=5), 7)
    plain = EpisodeWrapper(MockEnv(obs_size, 2, max_steps=5), 7)
    plain.env.step_deferred = None  # the env steps by itself
    k = keys.split(keys.key(3, dev), n)
    sa, sb = fused.reset(k), plain.reset(k)
    seen_done = seen_trunc = False
    for i in range(12):
        ra, rb = fused.reset(keys.fold_in(k, i + 100)), plain.reset(keys.fold_in(k, i + 100))
        with _lib.profiler as prof:
            stepped_a, after_a = fused.step_and_reset(sa, None, ra)
        assert [r[0] for r in prof.records] == ["mi_mock_episode_step_select"]
        with _lib.profiler as prof:
            stepped_b, after_b = plain.step_and_reset(sb, None, rb)
        assert [r[0] for r in prof.records] == ["mi_mock_env_step", "mi_episode_step_select"]
        for x, y in ((stepped_a, stepped_b), (after_a, after_b)):
            la, lb = tree_leaves(x), tree_leaves(y)
            assert len(la) == len(lb)
            for a, b in zip(la, lb):
                assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b)
        seen_done |= bool((stepped_a.done != 0).any())
        seen_trunc |= bool(stepped_a.info["truncated"].any())
        sa, sb = after_a, after_b
    assert seen_done and seen_trunc


@pytest.mark.parametrize("seed", [0, 17, 2**63 + 12345])
def test_key_kernels_equal_the_independent_restatement(dev, seed):
    """csrc/keys.hip against `oracle/keys.py` — the scheme restated in numpy uint64 with none
    of the product's code and pinned by the published SplitMix64 outputs
    (tests/test_oracle_keys.py): every key function the iteration uses (`ppo.py:271,284-294`,
    `rollout.py:57-59`, the synthetic envs' observation noise), bit for bit."""
    from oracle import keys as ok

    k = ok.key(seed)
    assert int(keys.key(seed, device=dev)) == int(k)
    kg = k.to(dev)
    assert torch.equal(keys.split(kg).cpu(), ok.split(k))
    kids = ok.split(k, (30, 257))                                     # [T, B] reset keys
    assert torch.equal(keys.split(kg, (30, 257)).cpu(), kids)
    kidg = kids.to(dev)
    a, b = keys.split2(kidg)
    assert torch.equal(torch.stack([a, b], -1).cpu(), ok.split(kids, 2))
    assert torch.equal(keys.split(kidg, 3).cpu(), ok.split(kids, 3))
    for data in (0, 3, 2**40 + 9):
        assert torch.equal(keys.fold_in(kidg, data).cpu(), ok.fold_in(kids, data))
    steps = (torch.arange(kids.numel(), dtype=torch.int64) * 977 - 3).reshape(kids.shape)
    assert torch.equal(keys.fold_key(kidg, steps.to(dev)).cpu(), ok.fold_key(kids, steps))
    assert torch.equal(keys.bits(kidg, (3,)).cpu(), ok.bits(kids, (3,)))
    assert torch.equal(keys.randint(kidg, (2,), -3, 10).cpu(), ok.randint(kids, (2,), -3, 10))
    assert torch.equal(keys.uniform(kidg, (5,)).cpu(), ok.uniform(kids, (5,)))
    assert torch.equal(keys.unit_uniform(kidg, (5,)).cpu(), ok.unit_uniform(kids, (5,)))
    assert torch.equal(keys.unit_uniform(kidg, (17,), fold=steps.to(dev)).cpu(),
                       ok.unit_uniform(kids, (17,), fold=steps))
    for n in (64, 1024, 4096, 8192):                                  # one launch for all 4
        assert torch.equal(keys.permutations(kg, 4, n).cpu(), ok.permutations(k, 4, n))
    assert torch.equal(keys.permutation(kg, 10000).cpu(), ok.permutation(k, 10000))
