"""Thin Python wrappers over the libmippo C ABI (one per entry point).

Each wrapper validates shapes/dtypes on the host (so a kernel never sees an
operand it was not sized for), allocates outputs with torch, and enqueues the
kernel on torch's current stream.  No wrapper synchronises, and none falls back
to torch arithmetic: a CPU tensor or a missing library raises `MippoError`.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from ._lib import MippoError, check, lib, profiler, ptr, stream

f32 = torch.float32
f64 = torch.float64
u8 = torch.uint8
i64 = torch.int64

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SWISH = 0, 1, 2, 3
ACT_CODES = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "tanh": ACT_TANH,
             "swish": ACT_SWISH}

# Workspaces are keyed by (device, stream, tag, bytes) and never freed or regrown,
# so a pointer baked into a captured HIP graph stays valid for the process
# lifetime, and kernels running concurrently on two streams never share one.
_workspaces: dict = {}


def workspace(device, tag: str, nbytes: int, zeroed: bool = False) -> torch.Tensor:
    """Per (device, stream, tag, size) scratch buffer.  `zeroed`: zero-filled when
    created (ticket counters; the kernels that use them leave them zero)."""
    key = (str(device), stream(), tag, int(nbytes))
    ws = _workspaces.get(key)
    if ws is None:
        alloc = torch.zeros if zeroed else torch.empty
        ws = alloc(max(int(nbytes), 16), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _as_u8(t: torch.Tensor) -> torch.Tensor:
    """bool tensors share storage layout with uint8 (0/1)."""
    if t.dtype == torch.bool:
        return t.view(torch.uint8)
    if t.dtype != torch.uint8:
        raise MippoError(f"flag tensor must be bool or uint8, got {t.dtype}")
    return t


def _need(cond: bool, msg: str) -> None:
    if not cond:
        raise MippoError(msg)


# ---------------------------------------------------------------- a13: GAE
def gae(rewards, values, last_value, done, truncated, gamma: float, lambda_: float,
        with_targets: bool = False, out=None, out_targets=None, with_stats: bool = False):
    """GAE reverse scan over `[T, N]` (reference ppo.py:351-394).
    Returns `advantages` or `(advantages, targets)` with `targets = V + A`;
    `with_stats` appends fp64 [3] = (sum A, sum A^2, count), reduced in the same launch."""
    _need(rewards.dim() == 2, f"rewards must be [T, N], got {tuple(rewards.shape)}")
    T, N = rewards.shape
    _need(values.shape == (T, N) and done.shape == (T, N) and truncated.shape == (T, N),
          "gae: values/done/truncated must match rewards [T, N]")
    _need(last_value.shape == (N,), f"gae: last_value must be [{N}]")
    done = _as_u8(done)
    truncated = _as_u8(truncated)
    adv = out if out is not None else torch.empty_like(rewards)
    tgt = None
    if with_targets:
        tgt = out_targets if out_targets is not None else torch.empty_like(rewards)
    if with_stats:
        _need(T >= 1 and N >= 1, "gae: with_stats needs a non-empty [T, N]")
        stats = torch.empty(3, dtype=f64, device=rewards.device)
        ws = workspace(rewards.device, "gae_stats", lib().mi_gae_stats_workspace_bytes(N),
                       zeroed=True)
        rc = lib().mi_gae_stats_f32(ptr(rewards, f32), ptr(values, f32), ptr(last_value, f32),
                                    ptr(done, u8), ptr(truncated, u8), ptr(adv, f32),
                                    ptr(tgt, f32), T, N, float(gamma), float(lambda_),
                                    ptr(stats, f64), ptr(ws), stream())
        check(rc, "mi_gae_stats_f32")
        return (adv, tgt, stats) if with_targets else (adv, stats)
    rc = lib().mi_gae_f32(ptr(rewards, f32), ptr(values, f32), ptr(last_value, f32),
                          ptr(done, u8), ptr(truncated, u8), ptr(adv, f32), ptr(tgt, f32),
                          T, N, float(gamma), float(lambda_), stream())
    check(rc, "mi_gae_f32")
    return (adv, tgt) if with_targets else adv


# ------------------------------------------------------- a8 / a16: normaliser
def normalize_fwd(x, mean, m2, counter, epsilon: float, out=None):
    F = mean.numel()
    _need(F >= 1 and tuple(x.shape[x.dim() - mean.dim():]) == tuple(mean.shape),
          "normalize_fwd: trailing dims of x must equal the statistics' shape")
    M = x.numel() // F
    out = out if out is not None else torch.empty_like(x)
    check(lib().mi_normalize_fwd_f32(ptr(x, f32), ptr(mean, f32), ptr(m2, f32), ptr(counter, f32),
                                     float(epsilon), ptr(out, f32), M, F, stream()),
          "mi_normalize_fwd_f32")
    return out


def normalize_fwd_tail(x, x_tail, mean, m2, counter, epsilon: float, out):
    """`normalize_fwd` of [x ; x_tail] into the contiguous `out` [M + M_tail, F] in one launch."""
    F = mean.numel()
    M, Mt = x.numel() // F, x_tail.numel() // F
    _need(out.is_contiguous() and out.numel() == (M + Mt) * F and x.is_contiguous()
          and x_tail.is_contiguous(), "normalize_fwd_tail: contiguous [M + M_tail, F] output")
    check(lib().mi_normalize_fwd_tail_f32(
        ptr(x, f32), ptr(x_tail, f32), ptr(mean, f32), ptr(m2, f32), ptr(counter, f32),
        float(epsilon), ptr(out, f32), M, Mt, F, stream()), "mi_normalize_fwd_tail_f32")
    return out


def normalize_bwd(g_out, m2, counter, epsilon: float):
    F = m2.numel()
    M = g_out.numel() // F
    g_x = torch.empty_like(g_out)
    check(lib().mi_normalize_bwd_f32(ptr(g_out, f32), ptr(m2, f32), ptr(counter, f32),
                                     float(epsilon), ptr(g_x, f32), M, F, stream()),
          "mi_normalize_bwd_f32")
    return g_x


def welford_batch_stats(x: torch.Tensor, F: int) -> torch.Tensor:
    """(n, mean, sum of squared deviations) per column of x viewed as [M, F];
    returns a [3, F] tensor."""
    _need(x.numel() % F == 0 and x.numel() > 0, "welford_batch_stats: bad shape")
    M = x.numel() // F
    nbytes = lib().mi_welford_workspace_bytes(M, F)
    _need(nbytes >= 0, "mi_welford_workspace_bytes failed")
    ws = workspace(x.device, "welford", nbytes)
    stats = torch.empty(3, F, dtype=f32, device=x.device)
    check(lib().mi_welford_batch_stats_f32(ptr(x, f32), ptr(stats, f32), ptr(ws), M, F, stream()),
          "mi_welford_batch_stats_f32")
    return stats


def welford_merge(mean, m2, counter, batch_stats, advance_counter: bool) -> None:
    F = mean.numel()
    _need(batch_stats.shape == (3, F), "welford_merge: batch_stats must be [3, F]")
    check(lib().mi_welford_merge_f32(ptr(mean, f32), ptr(m2, f32), ptr(counter, f32),
                                     ptr(batch_stats, f32), F, int(bool(advance_counter)),
                                     stream()), "mi_welford_merge_f32")


def col_mean_std(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """[2, C] = (mean, population std) of every column of a small [R, C] fp32 matrix."""
    _need(x.dim() == 2 and x.dtype == f32 and x.shape[0] >= 1 and x.is_contiguous(),
          "col_mean_std: x must be a contiguous [R, C] f32 matrix")
    R, C = x.shape
    out = torch.empty(2, C, dtype=f32, device=x.device)
    check(lib().mi_col_mean_std_f32(ptr(x, f32), R, C, float(scale), ptr(out, f32), stream()),
          "mi_col_mean_std_f32")
    return out


# ----------------------------------------------------------- a10: sampler
def tanh_gauss_fwd(mean_and_std, extras, rng_state, offset_add: int, *, min_std: float,
                   std_scale: float, entropy_weight: float, deterministic: bool,
                   eps=None, eps2=None, want_action=True, want_raw=True, want_reg=True,
                   want_stats=False):
    """Returns dict(raw, action, log_likelihood, reg, mu, sigma) (None where not asked)."""
    _need(mean_and_std.dim() == 2 and mean_and_std.shape[1] % 2 == 0,
          "tanh_gauss_fwd: mean_and_std must be [B, 2A]")
    B, A2 = mean_and_std.shape
    A = A2 // 2
    dev = mean_and_std.device
    if extras is not None:
        _need(extras.shape == (B, A), "tanh_gauss_fwd: extras must be [B, A]")
    for e in (eps, eps2):
        if e is not None:
            _need(e.shape == (B, A), "tanh_gauss_fwd: injected noise must be [B, A]")
    mk = lambda *s: torch.empty(*s, dtype=f32, device=dev)
    raw = mk(B, A) if (want_raw and extras is None) else None
    action = mk(B, A) if want_action else None
    ll = mk(B)
    reg = mk(B) if want_reg else None
    mu = mk(B, A) if want_stats else None
    sigma = mk(B, A) if want_stats else None
    check(lib().mi_tanh_gauss_fwd_f32(
        ptr(mean_and_std, f32), ptr(extras, f32), ptr(rng_state), int(offset_add),
        ptr(eps, f32), ptr(eps2, f32), ptr(raw, f32), ptr(action, f32), ptr(ll, f32),
        ptr(reg, f32), ptr(mu, f32), ptr(sigma, f32), B, A, float(min_std), float(std_scale),
        float(entropy_weight), int(bool(deterministic)), stream()), "mi_tanh_gauss_fwd_f32")
    return dict(raw=raw if extras is None else extras, action=action, log_likelihood=ll,
                reg=reg, mu=mu, sigma=sigma)


def tanh_gauss_bwd(mean_and_std, extras, rng_state, offset_add: int, g_ll, g_reg: float, *,
                   min_std: float, std_scale: float, entropy_weight: float, eps2=None):
    B, A2 = mean_and_std.shape
    A = A2 // 2
    _need(extras.shape == (B, A), "tanh_gauss_bwd: extras must be [B, A]")
    if g_ll is not None:
        _need(g_ll.shape == (B,), "tanh_gauss_bwd: g_ll must be [B]")
    g = torch.empty_like(mean_and_std)
    check(lib().mi_tanh_gauss_bwd_f32(
        ptr(mean_and_std, f32), ptr(extras, f32), ptr(rng_state), int(offset_add),
        ptr(eps2, f32), ptr(g_ll, f32), float(g_reg), ptr(g, f32), B, A, float(min_std),
        float(std_scale), float(entropy_weight), stream()), "mi_tanh_gauss_bwd_f32")
    return g


def philox_normal(rng_state, offset_add: int, n: int):
    dev = rng_state.device
    eps = torch.empty(n, dtype=f32, device=dev)
    eps2 = torch.empty(n, dtype=f32, device=dev)
    check(lib().mi_philox_normal_f32(ptr(rng_state), int(offset_add), ptr(eps, f32),
                                     ptr(eps2, f32), n, stream()), "mi_philox_normal_f32")
    return eps, eps2


def rng_advance(rng_state, n: int) -> None:
    check(lib().mi_rng_advance(ptr(rng_state), int(n), stream()), "mi_rng_advance")


def make_rng_state(seed: int, device, offset: int = 0) -> torch.Tensor:
    """Device-resident {seed, offset} pair (uint64 stored as int64 bit patterns)."""
    def s64(v):
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >= (1 << 63) else v
    return torch.tensor([s64(int(seed)), s64(int(offset))], dtype=i64, device=device)


# -------------------------------------------------------------- a9: dense
def dense_fwd(x, w, bias, act: int, want_preact: bool = False):
    _need(x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[0],
          f"dense_fwd: x {tuple(x.shape)} @ w {tuple(w.shape)}")
    M, K = x.shape
    N = w.shape[1]
    if bias is not None:
        _need(bias.shape == (N,), "dense_fwd: bias must be [N]")
    y = torch.empty(M, N, dtype=f32, device=x.device)
    pre = torch.empty(M, N, dtype=f32, device=x.device) if want_preact else None
    check(lib().mi_dense_fwd_f32(ptr(x, f32), ptr(w, f32), ptr(bias, f32), ptr(y, f32),
                                 ptr(pre, f32), M, K, N, int(act), stream()), "mi_dense_fwd_f32")
    return (y, pre) if want_preact else y


def dense_bwd_dx(g_y, aux, w, act: int):
    M, N = g_y.shape
    K = w.shape[0]
    _need(w.shape == (K, N), "dense_bwd_dx: w must be [K, N]")
    if act != ACT_NONE:
        _need(aux is not None and aux.shape == (M, N), "dense_bwd_dx: aux must be [M, N]")
    g_x = torch.empty(M, K, dtype=f32, device=g_y.device)
    check(lib().mi_dense_bwd_dx_f32(ptr(g_y, f32), ptr(aux, f32) if act != ACT_NONE else None,
                                    ptr(w, f32), ptr(g_x, f32), M, K, N, int(act), stream()),
          "mi_dense_bwd_dx_f32")
    return g_x


def dense_bwd_dw(x, g_y, aux, g_w, g_b, act: int, accumulate: bool = True) -> None:
    M, K = x.shape
    N = g_y.shape[1]
    _need(g_y.shape == (M, N) and g_w.shape == (K, N), "dense_bwd_dw: shape mismatch")
    if g_b is not None:
        _need(g_b.shape == (N,), "dense_bwd_dw: g_b must be [N]")
    if act != ACT_NONE:
        _need(aux is not None and aux.shape == (M, N), "dense_bwd_dw: aux must be [M, N]")
    nbytes = lib().mi_dense_bwd_dw_workspace_bytes(M, K, N)
    _need(nbytes >= 0, "mi_dense_bwd_dw_workspace_bytes failed")
    ws = workspace(x.device, "dense_dw", nbytes)
    check(lib().mi_dense_bwd_dw_f32(ptr(x, f32), ptr(g_y, f32),
                                    ptr(aux, f32) if act != ACT_NONE else None, ptr(g_w, f32),
                                    ptr(g_b, f32), ptr(ws), M, K, N, int(act),
                                    int(bool(accumulate)), stream()), "mi_dense_bwd_dw_f32")


# ------------------------------------------------------- a9: dense, bf16 MFMA
bf16 = torch.bfloat16


def pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def _bf_buf(rows: int, cols: int, device) -> torch.Tensor:
    """bf16 [rows][pad8(cols)] operand buffer (producers write the zero padding)."""
    return torch.empty(rows, pad8(cols), dtype=bf16, device=device)


def cast_pad_bf16(x: torch.Tensor, aux=None, act: int = ACT_NONE) -> torch.Tensor:
    """fp32 [M, F] (x act'(aux)) -> bf16 [M, pad8(F)], zero padded."""
    M, F = x.shape
    out = _bf_buf(M, F, x.device)
    check(lib().mi_cast_pad_bf16(ptr(x, f32), ptr(aux), aux.shape[1] if aux is not None else 0,
                                 int(act), ptr(out), out.shape[1], M, F, stream()),
          "mi_cast_pad_bf16")
    return out


def weights_to_bf16(w: torch.Tensor, w_bf: torch.Tensor, wt_bf: torch.Tensor) -> None:
    K, N = w.shape
    _need(w_bf.shape == (K, pad8(N)) and wt_bf.shape == (N, pad8(K)), "weights_to_bf16: shapes")
    check(lib().mi_weights_to_bf16(ptr(w, f32), ptr(w_bf, bf16), w_bf.shape[1], ptr(wt_bf, bf16),
                                   wt_bf.shape[1], K, N, stream()), "mi_weights_to_bf16")


def frag_sizes(K: int, N: int):
    """Element counts of the (forward, backward) fragment-major images of a [K, N] layer."""
    return (((N + 15) // 16) * ((K + 31) // 32) * 512, ((K + 15) // 16) * ((N + 31) // 32) * 512)


def weights_to_bf16_multi(ws: list, w_bfs: list, wt_bfs: list, ffs=None, fbs=None) -> None:
    """Refresh the bf16 shadows (row-major W, W^T and, if given, the two
    fragment-major images) of up to 16 layers per launch."""
    for i in range(0, len(ws), 16):
        w, wb, wt = ws[i:i + 16], w_bfs[i:i + 16], wt_bfs[i:i + 16]
        n = len(w)
        P = ctypes.c_void_p * n
        I = ctypes.c_int64 * n
        ff = P(*[ptr(t, bf16) for t in ffs[i:i + 16]]) if ffs is not None else None
        fb = P(*[ptr(t, bf16) for t in fbs[i:i + 16]]) if fbs is not None else None
        check(lib().mi_weights_to_bf16_multi(
            n, P(*[ptr(t, f32) for t in w]), P(*[ptr(t, bf16) for t in wb]),
            P(*[ptr(t, bf16) for t in wt]), ff, fb, I(*[t.shape[0] for t in w]),
            I(*[t.shape[1] for t in w]), stream()), "mi_weights_to_bf16_multi")


def dense_fwd_bf16(x_bf, wt_bf, bias, K: int, N: int, act: int, *, want_f32: bool,
                   want_bf: bool, want_preact: bool = False):
    """Returns (y_f32 | None, y_bf | None, preact_bf | None)."""
    M = x_bf.shape[0]
    _need(x_bf.dtype == bf16 and x_bf.shape[1] == pad8(K), "dense_fwd_bf16: x_bf must be [M, pad8(K)]")
    _need(wt_bf.shape == (N, pad8(K)), "dense_fwd_bf16: wt_bf must be [N, pad8(K)]")
    dev = x_bf.device
    y_f32 = torch.empty(M, N, dtype=f32, device=dev) if want_f32 else None
    y_bf = _bf_buf(M, N, dev) if want_bf else None
    pre = _bf_buf(M, N, dev) if want_preact else None
    check(lib().mi_dense_fwd_bf16(
        ptr(x_bf, bf16), x_bf.shape[1], ptr(wt_bf, bf16), wt_bf.shape[1], ptr(bias, f32),
        ptr(y_f32, f32), ptr(y_bf), pad8(N), ptr(pre), M, K, N, int(act), stream()),
        "mi_dense_fwd_bf16")
    return y_f32, y_bf, pre


def dense_bwd_dx_bf16(dz_bf, w_bf, prev_bf, prev_act: int, K: int, N: int, *, want_f32: bool,
                      want_bf: bool):
    """Returns (gx_f32 | None, gx_bf | None)."""
    M = dz_bf.shape[0]
    _need(dz_bf.shape[1] == pad8(N) and w_bf.shape == (K, pad8(N)), "dense_bwd_dx_bf16: shapes")
    dev = dz_bf.device
    gx_f32 = torch.empty(M, K, dtype=f32, device=dev) if want_f32 else None
    gx_bf = _bf_buf(M, K, dev) if want_bf else None
    if prev_act != ACT_NONE:
        _need(prev_bf is not None and prev_bf.shape == (M, pad8(K)), "dense_bwd_dx_bf16: prev")
    check(lib().mi_dense_bwd_dx_bf16(
        ptr(dz_bf, bf16), dz_bf.shape[1], ptr(w_bf, bf16), w_bf.shape[1],
        ptr(prev_bf) if prev_act != ACT_NONE else None, pad8(K), int(prev_act), ptr(gx_f32, f32),
        ptr(gx_bf), pad8(K), M, K, N, stream()), "mi_dense_bwd_dx_bf16")
    return gx_f32, gx_bf


def dense_bwd_dw_bf16(x_bf, dz_bf, g_w, g_b, accumulate: bool = True) -> None:
    K, N = g_w.shape
    M = x_bf.shape[0]
    _need(x_bf.shape == (M, pad8(K)) and dz_bf.shape == (M, pad8(N)),
          "dense_bwd_dw_bf16: operands must be [M, pad8(K)] / [M, pad8(N)]")
    nbytes = lib().mi_dense_bwd_dw_bf16_workspace_bytes(M, K, N)
    _need(nbytes >= 0, "mi_dense_bwd_dw_bf16_workspace_bytes failed")
    ws = workspace(g_w.device, "dense_dw_bf16", nbytes)
    check(lib().mi_dense_bwd_dw_bf16(ptr(x_bf, bf16), x_bf.shape[1], ptr(dz_bf, bf16),
                                     dz_bf.shape[1], ptr(g_w, f32), ptr(g_b, f32), ptr(ws), M, K,
                                     N, int(bool(accumulate)), stream()), "mi_dense_bwd_dw_bf16")


class _SlabDefer:
    """Deferred weight gradients of one gradient step.  `Optimizer.begin(defer_dw=True)`
    opens it (the training loops): every grouped dW request whose gradients live in the
    optimiser's arena is QUEUED instead of launched, so that at `Optimizer.update` the dW /
    db of every layer of the network — whatever module asked for them — go out as ONE
    grouped launch per 8 problems (a Dense-GRU-Dense actor beside an MLP critic is five
    requests of 1-3 layers), and the split-M slabs of the last one are summed inside the
    Adam launch (`mi_adam_step_slabs_f32`) instead of by their own reduction launch.
    Anything that reads the gradients first (norm, all-reduce) calls
    `flush_pending_slabs`, which launches and reduces everything.  Off by default:
    `.grad` is complete after every backward."""
    arena = None    # float32 [n] while a deferring gradient step is open
    queue: list = []   # (x_bf, dz_bf, g_w, g_b, gw_off, gb_off, b_lo) not yet launched
    pending = None  # (slab_ptrs, S, Ks, Ns, g_w, g_b, gw_off, gb_off, workspace, b_lo):
    #                 launched, slabs not yet reduced

    def offsets(self, g_w, g_b, K, N, b_lo=0):
        """Arena offsets of a problem's gradients, or None if one lies outside the arena
        (`b_lo` > 0: g_b holds bias columns b_lo .. N-1 only)."""
        if self.arena is None:
            return None
        base, n = self.arena.data_ptr(), self.arena.numel()
        ow = (g_w.data_ptr() - base) // 4
        ob = -1 if g_b is None else (g_b.data_ptr() - base) // 4
        if not (g_w.is_contiguous() and 0 <= ow and ow + K * N <= n):
            return None
        if g_b is not None and not (g_b.is_contiguous() and g_b.numel() == N - b_lo
                                    and 0 <= ob and ob + N - b_lo <= n):
            return None
        return ow, ob


slab_defer = _SlabDefer()


def _dw_group(grp: list, accumulate: bool, fold) -> None:
    """One grouped dW launch of <= 8 problems sharing M.  `fold` = [(gw_off, gb_off, b_lo)]
    per problem: leave the slabs pending for the optimiser launch instead of reducing them.
    A problem may carry a 5th element b_lo > 0 (its g_b holds bias columns b_lo .. N-1 only):
    folded, the optimiser launch picks those columns out of the slabs; reduced here, the N
    column sums go to a scratch vector whose tail is then added to g_b."""
    n = len(grp)
    M = grp[0][0].shape[0]
    tails = []
    if fold is None:
        full = []
        for pr in grp:
            b_lo = pr[4] if len(pr) > 4 else 0
            if b_lo and pr[3] is not None:
                tmp = torch.zeros(pr[2].shape[1], dtype=f32, device=pr[2].device)
                tails.append((pr[3], tmp, b_lo))
                pr = (pr[0], pr[1], pr[2], tmp)
            full.append(tuple(pr[:4]))
        grp = full
    else:
        grp = [tuple(pr[:4]) for pr in grp]
    Ks = [g[2].shape[0] for g in grp]
    Ns = [g[2].shape[1] for g in grp]
    for (x_bf, dz_bf, g_w, g_b), K, N in zip(grp, Ks, Ns):
        _need(x_bf.shape == (M, pad8(K)) and dz_bf.shape == (M, pad8(N)),
              "dense_bwd_dw_grouped_bf16: operands must be [M, pad8(K)] / [M, pad8(N)], got "
              f"{tuple(x_bf.shape)} / {tuple(dz_bf.shape)} for M={M} K={K} N={N}")
    P = ctypes.c_void_p * n
    I = ctypes.c_int64 * n
    Kc, Nc = I(*Ks), I(*Ns)
    nbytes = lib().mi_dense_bwd_dw_grouped_bf16_workspace_bytes(n, Kc, Nc, M)
    _need(nbytes >= 0, "mi_dense_bwd_dw_grouped_bf16_workspace_bytes failed")
    _reduce_pending()  # the slabs of a pending launch live in the same workspace
    ws = workspace(grp[0][2].device, "dense_dw_grouped", nbytes)
    if profiler.active:
        profiler.next_flops = 2.0 * M * sum(K * N for K, N in zip(Ks, Ns))
        # operands once + fp32 gradients read-modify-written
        profiler.next_bytes = (sum((g[0].numel() + g[1].numel()) * 2.0 for g in grp)
                               + sum(8.0 * (K * N + N) for K, N in zip(Ks, Ns)))
    xs = P(*[ptr(g[0], bf16) for g in grp])
    dzs = P(*[ptr(g[1], bf16) for g in grp])
    if fold is not None:
        sp, S = P(), I()
        check(lib().mi_dense_bwd_dw_grouped_slabs_bf16(n, xs, dzs, Kc, Nc, M, ptr(ws), sp, S,
                                                       stream()),
              "mi_dense_bwd_dw_grouped_slabs_bf16")
        slab_defer.pending = (list(sp), list(S), Ks, Ns, [g[2] for g in grp],
                              [g[3] for g in grp], [o[0] for o in fold],
                              [o[1] for o in fold], ws, [o[2] for o in fold])
        return
    check(lib().mi_dense_bwd_dw_grouped_bf16(
        n, xs, dzs, P(*[ptr(g[2], f32) for g in grp]), P(*[ptr(g[3], f32) for g in grp]),
        Kc, Nc, M, ptr(ws), int(bool(accumulate)), stream()), "mi_dense_bwd_dw_grouped_bf16")
    for g_b, tmp, b_lo in tails:
        g_b += tmp[b_lo:]


def _reduce_pending() -> None:
    pend, slab_defer.pending = slab_defer.pending, None
    if pend is None:
        return
    sp, S, Ks, Ns, g_w, g_b = pend[:6]
    b_lo = pend[9]
    n = len(Ks)
    P = ctypes.c_void_p * n
    I = ctypes.c_int64 * n
    tails = []
    g_b = list(g_b)
    for l in range(n):  # a bias tail: the N column sums to scratch, its tail added below
        if b_lo[l] and g_b[l] is not None:
            tmp = torch.zeros(Ns[l], dtype=f32, device=g_w[l].device)
            tails.append((g_b[l], tmp, b_lo[l]))
            g_b[l] = tmp
    check(lib().mi_reduce_slabs_grouped_f32(
        n, P(*sp), I(*S), I(*Ks), I(*Ns), P(*[ptr(t, f32) for t in g_w]),
        P(*[ptr(t, f32) for t in g_b]), 1, stream()), "mi_reduce_slabs_grouped_f32")
    for dst, tmp, lo in tails:
        dst += tmp[lo:]


def _run_queue(fold_last: bool) -> None:
    """Launch the queued problems: grouped by M (request order kept), 8 per launch; with
    `fold_last` the final launch leaves its slabs pending."""
    q, slab_defer.queue = slab_defer.queue, []
    if not q:
        return
    by_m: dict = {}
    for pr in q:
        by_m.setdefault(pr[0].shape[0], []).append(pr)
    groups = [g[i:i + 8] for g in by_m.values() for i in range(0, len(g), 8)]
    for gi, grp in enumerate(groups):
        last = fold_last and gi == len(groups) - 1
        _dw_group([(*pr[:4], pr[6]) for pr in grp], True,
                  [(pr[4], pr[5], pr[6]) for pr in grp] if last else None)


def flush_pending_slabs() -> None:
    """Launch every queued dW request and reduce every pending slab now: after this the
    gradients are complete."""
    _run_queue(fold_last=False)
    _reduce_pending()


def take_pending_slabs():
    """The optimiser is about to run: launch the queue, the last launch's slabs (if any)
    are handed to it unreduced."""
    _run_queue(fold_last=True)
    pend, slab_defer.pending = slab_defer.pending, None
    return pend


def _dw_tiles(K: int, N: int) -> int:
    """Output tiles of one dW problem (csrc/gemm_bf16.hip: 128 x {128, 64, 16})."""
    tn = 128 if N > 64 else (64 if N > 16 else 16)
    return -(-K // 128) * -(-N // tn)


# A grouped launch splits M so that (tiles x splits) fills one resident wave of workgroups
# (~416).  Small requests (a few tiles: launch-bound) gain from sharing a launch; past ~26
# tiles the shared launch has too few splits to keep the chip busy (measured at C3:
# 4x256 + 2x512 trunks in one launch 14.5 ms / iteration against 12.6 ms as two), so
# requests are queued only while the queue stays under this many tiles — and only requests of
# at most half that: a big request launched at once overlaps the other branch of a
# two-stream backward (networks/adapter.py), which the queue would serialise.
_DW_QUEUE_TILES = 26
_DW_REQUEST_TILES = 13


def dense_bwd_dw_grouped_bf16(problems: list, accumulate: bool = True,
                              bias_first: list | None = None) -> None:
    """dW / db of several layers that share M: `problems` = [(x_bf, dz_bf, g_w, g_b)].
    `bias_first[i]` = b_lo > 0: problem i's g_b is [N - b_lo] and receives the column sums
    of dz columns b_lo .. N-1 only (a GRU's recurrent kernel: the n gate's bias)."""
    bias_first = [0] * len(problems) if bias_first is None else [int(b) for b in bias_first]
    _need(len(bias_first) == len(problems), "dense_bwd_dw_grouped_bf16: one bias_first each")
    _need(accumulate or not any(bias_first), "dense_bwd_dw_grouped_bf16: bias tails accumulate")
    tiles = sum(_dw_tiles(g_w.shape[0], g_w.shape[1]) for _, _, g_w, _ in problems)
    queued = sum(_dw_tiles(pr[2].shape[0], pr[2].shape[1]) for pr in slab_defer.queue)
    if (accumulate and slab_defer.arena is not None and tiles <= _DW_REQUEST_TILES
            and tiles + queued <= _DW_QUEUE_TILES):
        offs = [slab_defer.offsets(g_w, g_b, g_w.shape[0], g_w.shape[1], lo)
                for (_, _, g_w, g_b), lo in zip(problems, bias_first)]
        if all(o is not None for o in offs):
            # one gradient twice in a launch would race in the slab reduction (weight
            # sharing): what is queued goes out first
            seen = {pr[2].data_ptr() for pr in slab_defer.queue}
            mine = [pr[2].data_ptr() for pr in problems]
            if len(set(mine)) < len(mine):
                offs = None
            elif seen & set(mine):
                _run_queue(fold_last=False)
            if offs is not None:
                slab_defer.queue.extend(
                    (x, dz, g_w, g_b, o[0], o[1], lo)
                    for (x, dz, g_w, g_b), o, lo in zip(problems, offs, bias_first))
                return
    # launched now; what is queued stays queued unless it shares a gradient with this request
    # (the order of the additions into one gradient is then kept)
    mine = {pr[2].data_ptr() for pr in problems} | {pr[3].data_ptr() for pr in problems
                                                   if pr[3] is not None}
    if any(pr[2].data_ptr() in mine or (pr[3] is not None and pr[3].data_ptr() in mine)
           for pr in slab_defer.queue):
        _run_queue(fold_last=False)
    full = [(*pr, lo) for pr, lo in zip(problems, bias_first)]
    for i in range(0, len(full), 8):
        _dw_group(full[i:i + 8], accumulate, None)


def mlp_fwd_bf16(x: torch.Tensor, wts: list, biases: list, dims: list, acts: list, *,
                 train: bool, want_out: bool = True):
    """Fused MLP trunk forward.  Returns (out_f32 [M, N_last], saved) where `saved`
    (training only) is a list per layer of (x_bf, aux_bf) — the bf16 input of the
    layer and the tensor its activation derivative is evaluated on."""
    M, K0 = x.shape
    L = len(wts)
    _need(len(dims) == L + 1 and dims[0] == K0 and len(acts) == L, "mlp_fwd_bf16: dims/acts")
    dev = x.device
    # `want_out=False` (training form whose last layer keeps its bf16 image): no fp32 output
    _need(want_out or (train and acts[-1] != ACT_NONE), "mlp_fwd_bf16: nothing would be written")
    out = torch.empty(M, dims[-1], dtype=f32, device=dev) if want_out else None
    P = ctypes.c_void_p * L
    I = ctypes.c_int64 * (L + 1)
    y_bf = [None] * L
    pre_bf = [None] * L
    x_bf = None
    if train:
        x_bf = _bf_buf(M, K0, dev)
        for l in range(L):
            last = l == L - 1
            N = dims[l + 1]
            # every image is written whole by the kernel, zero padding included
            if not last or acts[l] != ACT_NONE:
                y_bf[l] = _bf_buf(M, N, dev)
            if acts[l] == ACT_SWISH:
                pre_bf[l] = _bf_buf(M, N, dev)
    arr = lambda ts: P(*[ptr(t) for t in ts])
    if profiler.active:
        profiler.next_flops = 2.0 * M * sum(dims[l] * dims[l + 1] for l in range(L))
        w_bytes = sum(2 * dims[l] * dims[l + 1] + 4 * dims[l + 1] for l in range(L))
        kept = sum(t.numel() * 2 for t in [x_bf, *y_bf, *pre_bf] if t is not None)
        # algorithmic HBM bytes: fp32 input + weights in, fp32 output + kept bf16 images out
        profiler.next_bytes = (4.0 * M * K0 + w_bytes + (4.0 * M * dims[-1] if want_out else 0)
                               + kept)
    check(lib().mi_mlp_fwd_bf16(
        ptr(x, f32), M, L, arr(wts), arr(biases), I(*[int(d) for d in dims]),
        (ctypes.c_int64 * L)(*[int(a) for a in acts]), ptr(out, f32),
        arr(y_bf) if train else None, arr(pre_bf) if train else None, ptr(x_bf), stream()),
        "mi_mlp_fwd_bf16")
    if not train:
        return out, None
    saved = []
    for l in range(L):
        x_l = x_bf if l == 0 else y_bf[l - 1]
        aux = pre_bf[l] if acts[l] == ACT_SWISH else y_bf[l]
        saved.append((x_l, aux))
    return out, saved


def mlp_ws_supported(dims: list, acts: list) -> bool:
    L = len(acts)
    return bool(lib().mi_mlp_ws_supported(L, (ctypes.c_int64 * (L + 1))(*[int(d) for d in dims]),
                                          (ctypes.c_int64 * L)(*[int(a) for a in acts])))


def mlp_ws_fwd_bf16(x: torch.Tensor, wts: list, biases: list, dims: list, acts: list, *,
                    train: bool):
    """`mlp_fwd_bf16` on the weights-stationary kernel (csrc/trunk_ws.hip): same operands,
    same results bit for bit; for trunks `mlp_ws_supported` accepts."""
    M, K0 = x.shape
    L = len(wts)
    _need(len(dims) == L + 1 and dims[0] == K0 and len(acts) == L, "mlp_ws_fwd_bf16: dims/acts")
    dev = x.device
    out = torch.empty(M, dims[-1], dtype=f32, device=dev)
    P = ctypes.c_void_p * L
    I = ctypes.c_int64 * (L + 1)
    y_bf = [None] * L
    x_bf = None
    if train:
        x_bf = _bf_buf(M, K0, dev)
        for l in range(L - 1):
            y_bf[l] = _bf_buf(M, dims[l + 1], dev)
    arr = lambda ts: P(*[ptr(t) for t in ts])
    if profiler.active:
        profiler.next_flops = 2.0 * M * sum(dims[l] * dims[l + 1] for l in range(L))
        w_bytes = sum(2 * dims[l] * dims[l + 1] + 4 * dims[l + 1] for l in range(L))
        kept = sum(t.numel() * 2 for t in [x_bf, *y_bf] if t is not None)
        profiler.next_bytes = 4.0 * M * K0 + w_bytes + 4.0 * M * dims[-1] + kept
    check(lib().mi_mlp_ws_fwd_bf16(
        ptr(x, f32), M, L, arr(wts), arr(biases), I(*[int(d) for d in dims]),
        (ctypes.c_int64 * L)(*[int(a) for a in acts]), ptr(out, f32),
        arr(y_bf) if train else None, ptr(x_bf), stream()), "mi_mlp_ws_fwd_bf16")
    if not train:
        return out, None
    saved = [((x_bf if l == 0 else y_bf[l - 1]), y_bf[l]) for l in range(L)]
    return out, saved


def policy_ws_supported(a_dims: list, a_acts: list, c_dims: list, c_acts: list) -> bool:
    """Both trunks in the shape class of the weights-stationary kernels (csrc/trunk_ws.hip)."""
    i64s = lambda v: (ctypes.c_int64 * len(v))(*[int(q) for q in v])
    return bool(lib().mi_policy_ws_supported(len(a_acts), i64s(a_dims), i64s(a_acts),
                                             len(c_acts), i64s(c_dims), i64s(c_acts)))


def policy_ws_dual_supported(a_dims: list, a_acts: list, c_dims: list, c_acts: list) -> bool:
    """The weights-stationary policy step takes these two trunks as ONE launch at rollout
    sizes (<= 8192 rows)."""
    i64s = lambda v: (ctypes.c_int64 * len(v))(*[int(q) for q in v])
    return bool(lib().mi_policy_ws_dual_supported(len(a_acts), i64s(a_dims), i64s(a_acts),
                                                  len(c_acts), i64s(c_dims), i64s(c_acts)))


def policy_fwd_bf16(obs: torch.Tensor, norm, actor, critic, rng_state, offset_add: int, *,
                    min_std: float, std_scale: float, entropy_weight: float,
                    deterministic: bool, extras=None, eps=None, eps2=None, train: bool = False,
                    want_stats: bool = True, value_tail=None, ws: bool = False):
    """Normaliser -> action trunk -> sampler and value trunk in ONE launch
    (`mi_policy_fwd_bf16`).  `norm` = (mean, m2, counter, eps) | None; `actor` / `critic`
    = (frag images, biases, dims, acts).  Returns a dict: raw, action, log_likelihood,
    reg, mu, sigma, value and — training — mean_and_std, actor_saved, critic_saved
    (per-layer (x_bf, aux_bf) as `mlp_fwd_bf16`).  `value_tail` [Mt, K0]: extra rows for
    the value trunk only (the bootstrap observation); `value` and the critic images then
    have M + Mt rows and `value_tail_out` is the view of the last Mt.  `ws`: the
    weights-stationary kernels (`mi_policy_ws_fwd_bf16`: training sizes for trunks that
    `policy_ws_supported` accepts — one launch per trunk; up to 8192 rows for pairs that
    `policy_ws_dual_supported` accepts — both trunks in one launch) — same results bit for
    bit."""
    M, K0 = obs.shape
    dev = obs.device
    (a_w, a_b, a_dims, a_acts), (c_w, c_b, c_dims, c_acts) = actor, critic
    La, Lc = len(a_w), len(c_w)
    _need(a_dims[0] == K0 and c_dims[0] == K0, "policy_fwd_bf16: trunk inputs must match obs")
    A2 = a_dims[-1]
    A = A2 // 2
    if extras is not None:
        _need(extras.shape == (M, A), "policy_fwd_bf16: extras must be [M, A]")
    mk = lambda *sh: torch.empty(*sh, dtype=f32, device=dev)
    replay = extras is not None
    raw = None if replay else mk(M, A)
    action = None if replay else mk(M, A)
    ll, reg = mk(M), mk(M)
    mu = mk(M, A) if want_stats else None
    sigma = mk(M, A) if want_stats else None
    Mt = 0 if value_tail is None else value_tail.shape[0]
    if value_tail is not None:
        _need(value_tail.shape == (Mt, K0) and value_tail.is_contiguous(),
              "policy_fwd_bf16: value_tail must be a contiguous [Mt, K0]")
    value = mk(M + Mt, c_dims[-1])
    ms = mk(M, A2) if (train or ws) else None  # the ws kernels always write the head's rows

    def images(L, dims, acts, rows):
        if not train:
            return None, None, None
        y, pre = [None] * L, [None] * L
        for l in range(L):
            if l < L - 1 or acts[l] != ACT_NONE:
                y[l] = _bf_buf(rows, dims[l + 1], dev)
            if acts[l] == ACT_SWISH:
                pre[l] = _bf_buf(rows, dims[l + 1], dev)
        return y, pre, _bf_buf(rows, K0, dev)

    a_y, a_pre, a_x = images(La, a_dims, a_acts, M)
    c_y, c_pre, c_x = images(Lc, c_dims, c_acts, M + Mt)
    a_mask = c_mask = None
    if ws and train:
        # relu trunks keep no pre-activations: the `pre` slots carry the relu' masks the
        # weights-stationary backward reads (4 bits per lane of a 16 x 16 tile, one byte)
        def masks(L, dims, rows):  # [64-row block][column tile][lane][row tile of the block]
            return [torch.empty(-(-rows // 64), dims[l + 1] // 16, 64, 4, dtype=torch.uint8,
                                device=dev) if l < L - 1 else None for l in range(L)]
        a_mask, c_mask = masks(La, a_dims, M), masks(Lc, c_dims, M + Mt)
        a_pre, c_pre = a_mask, c_mask
    arr = lambda ts, L: None if ts is None else (ctypes.c_void_p * L)(*[ptr(t) for t in ts])
    i64s = lambda v: (ctypes.c_int64 * len(v))(*[int(q) for q in v])
    n_mean, n_m2, n_cnt, n_eps = norm if norm is not None else (None, None, None, 0.0)
    if profiler.active:
        flop = sum(a_dims[l] * a_dims[l + 1] for l in range(La)) \
            + sum(c_dims[l] * c_dims[l + 1] for l in range(Lc))
        profiler.next_flops = 2.0 * M * flop
        kept = sum(t.numel() * t.element_size()
                   for t in [a_x, c_x, *(a_y or []), *(a_pre or []), *(c_y or []), *(c_pre or [])]
                   if t is not None)
        w_bytes = sum(2 * a_dims[l] * a_dims[l + 1] for l in range(La)) \
            + sum(2 * c_dims[l] * c_dims[l + 1] for l in range(Lc))
        outs = sum(t.numel() * 4 for t in [raw, action, ll, reg, mu, sigma, value, ms, extras]
                   if t is not None)
        profiler.next_bytes = 4.0 * M * K0 + w_bytes + kept + outs
    if ws:
        _need(train or M + Mt <= 8192, "policy_fwd_bf16: without `train` the weights-stationary "
              "form is the one-launch rollout step (<= 8192 rows)")
    entry = lib().mi_policy_ws_fwd_bf16 if ws else lib().mi_policy_fwd_bf16
    check(entry(
        ptr(obs, f32), M, ptr(n_mean, f32), ptr(n_m2, f32), ptr(n_cnt, f32), float(n_eps),
        La, arr(a_w, La), arr(a_b, La), i64s(a_dims), i64s(a_acts),
        Lc, arr(c_w, Lc), arr(c_b, Lc), i64s(c_dims), i64s(c_acts),
        ptr(extras, f32), ptr(rng_state), int(offset_add), ptr(eps, f32), ptr(eps2, f32),
        float(min_std), float(std_scale), float(entropy_weight), int(bool(deterministic)),
        ptr(ms, f32), ptr(raw, f32), ptr(action, f32), ptr(ll, f32), ptr(reg, f32),
        ptr(mu, f32), ptr(sigma, f32), ptr(value, f32),
        arr(a_y, La), arr(a_pre, La), ptr(a_x), arr(c_y, Lc), arr(c_pre, Lc), ptr(c_x),
        ptr(value_tail, f32), Mt, stream()),
        "mi_policy_ws_fwd_bf16" if ws else "mi_policy_fwd_bf16")
    out = dict(raw=extras if replay else raw, action=action, log_likelihood=ll, reg=reg, mu=mu,
               sigma=sigma, value=value[:M], value_tail_out=value[M:] if Mt else None,
               mean_and_std=ms, actor_masks=a_mask, critic_masks=c_mask)
    if a_mask is not None:
        a_pre, c_pre = [None] * La, [None] * Lc  # (relu: no pre-activation images)
    if train:
        def saved(L, acts, x_bf, y, pre):
            return [((x_bf if l == 0 else y[l - 1]), (pre[l] if acts[l] == ACT_SWISH else y[l]))
                    for l in range(L)]
        out["actor_saved"] = saved(La, a_acts, a_x, a_y, a_pre)
        # the backward and the dW run over the first M rows of the critic images
        head = lambda t: None if t is None else t[:M]
        out["critic_saved"] = [(head(xb), head(aux))
                               for xb, aux in saved(Lc, c_acts, c_x, c_y, c_pre)]
    return out


def gru_policy_step_supported(K0: int, H: int, A2: int, c_dims: list, c_acts: list) -> bool:
    i64s = lambda v: (ctypes.c_int64 * len(v))(*[int(q) for q in v])
    return bool(lib().mi_gru_policy_step_supported(int(K0), int(H), int(A2), len(c_acts),
                                                   i64s(c_dims), i64s(c_acts)))


def gru_policy_step(obs: torch.Tensor, norm, dense_in, proj, w_h: torch.Tensor,
                    b_hn: torch.Tensor, dense_out, h_in: torch.Tensor, critic, rng_state,
                    offset_add: int, *, min_std: float, std_scale: float,
                    entropy_weight: float, deterministic: bool, eps=None, eps2=None):
    """One rollout step of the recurrent actor-critic in ONE launch
    (`mi_gru_policy_step_bf16`).  `dense_in` / `proj` / `dense_out` = (forward fragment
    image, bias); `critic` = (frag images, biases, dims, acts).  Returns a dict: h, raw,
    action, log_likelihood, reg, mu, sigma, value."""
    M, K0 = obs.shape
    H = h_in.shape[1]
    dev = obs.device
    c_w, c_b, c_dims, c_acts = critic
    Lc = len(c_w)
    A2 = dense_out[1].shape[0] if dense_out[1] is not None else None
    _need(A2 is not None, "gru_policy_step: the output layer needs a bias vector (its width)")
    A = A2 // 2
    _need(h_in.shape == (M, H) and h_in.is_contiguous() and w_h.shape == (H, 3 * H),
          "gru_policy_step: carry [M, H] and recurrent kernel [H, 3H]")
    mk = lambda *sh: torch.empty(*sh, dtype=f32, device=dev)
    h_out, raw, action = mk(M, H), mk(M, A), mk(M, A)
    ll, reg, mu, sigma = mk(M), mk(M), mk(M, A), mk(M, A)
    value = mk(M, c_dims[-1])
    P = ctypes.c_void_p * Lc
    i64s = lambda v: (ctypes.c_int64 * len(v))(*[int(q) for q in v])
    n_mean, n_m2, n_cnt, n_eps = norm if norm is not None else (None, None, None, 0.0)
    if profiler.active:
        flop = K0 * H + 6 * H * H + H * A2 + sum(c_dims[l] * c_dims[l + 1] for l in range(Lc))
        profiler.next_flops = 2.0 * M * flop
        profiler.next_bytes = 4.0 * M * (K0 + 2 * H + 4 * A + 3)
    check(lib().mi_gru_policy_step_bf16(
        ptr(obs, f32), M, K0, H, A2, ptr(n_mean, f32), ptr(n_m2, f32), ptr(n_cnt, f32),
        float(n_eps), ptr(dense_in[0], bf16), ptr(dense_in[1], f32), ptr(proj[0], bf16),
        ptr(proj[1], f32), ptr(w_h, f32), ptr(b_hn, f32), ptr(dense_out[0], bf16),
        ptr(dense_out[1], f32), ptr(h_in, f32), ptr(h_out, f32),
        Lc, P(*[ptr(t) for t in c_w]), P(*[ptr(t) for t in c_b]), i64s(c_dims), i64s(c_acts),
        ptr(rng_state), int(offset_add), ptr(eps, f32), ptr(eps2, f32), float(min_std),
        float(std_scale), float(entropy_weight), int(bool(deterministic)), None,
        ptr(raw, f32), ptr(action, f32), ptr(ll, f32), ptr(reg, f32), ptr(mu, f32),
        ptr(sigma, f32), ptr(value, f32), stream()), "mi_gru_policy_step_bf16")
    return dict(h=h_out, raw=raw, action=action, log_likelihood=ll, reg=reg, mu=mu, sigma=sigma,
                value=value)


def policy_bwd_bf16(mean_and_std, extras, rng_state, offset_add: int, g_ll, g_reg: float,
                    g_value, actor, critic, *, min_std: float, std_scale: float,
                    entropy_weight: float, eps2=None, ws: bool = False, masks=None):
    """Sampler backward + both dX chains in ONE launch (`mi_policy_bwd_bf16`).
    `actor` / `critic` = (backward frag images, dims, acts, auxs per layer).  Returns
    (actor dz list, critic dz list): bf16 [M, pad8(N_l)] per layer.  `ws`: the
    weights-stationary kernels (`mi_policy_ws_bwd_bf16`), same results bit for bit; `masks`
    = (actor masks, critic masks) of the weights-stationary forward (`policy_fwd_bf16(...,
    ws=True)["actor_masks" / "critic_masks"]`): relu' is read from them instead of the bf16
    images."""
    M, A2 = mean_and_std.shape
    dev = mean_and_std.device
    (a_w, a_dims, a_acts, a_aux), (c_w, c_dims, c_acts, c_aux) = actor, critic
    La, Lc = len(a_w), len(c_w)
    _need(extras.shape == (M, A2 // 2) and g_value.shape == (M, c_dims[-1]),
          "policy_bwd_bf16: shapes")
    if g_ll is not None:
        _need(g_ll.shape == (M,), "policy_bwd_bf16: g_ll must be [M]")
    a_dz = [_bf_buf(M, a_dims[l + 1], dev) for l in range(La)]
    c_dz = [_bf_buf(M, c_dims[l + 1], dev) for l in range(Lc)]
    arr = lambda ts, n: (ctypes.c_void_p * max(n, 1))(*[ptr(t) for t in ts])
    i64s = lambda v: (ctypes.c_int64 * len(v))(*[int(q) for q in v])
    if profiler.active:
        flop = sum(a_dims[l] * a_dims[l + 1] for l in range(1, La)) \
            + sum(c_dims[l] * c_dims[l + 1] for l in range(1, Lc))
        profiler.next_flops = 2.0 * M * flop
        # relu' operands: the bf16 images, or (weights-stationary, masks) a byte per 4 elements
        read = [*masks[0], *masks[1]] if (ws and masks is not None) \
            else [*a_aux[:La - 1], *c_aux[:Lc - 1]]
        moved = sum(t.numel() * t.element_size() for t in [*read, *a_dz, *c_dz]
                    if t is not None)
        w_bytes = sum(2 * a_dims[l] * a_dims[l + 1] for l in range(1, La)) \
            + sum(2 * c_dims[l] * c_dims[l + 1] for l in range(1, Lc))
        profiler.next_bytes = (4.0 * M * (A2 + A2 // 2 + 1 + c_dims[-1]) + w_bytes + moved)
    entry = lib().mi_policy_ws_bwd_bf16 if ws else lib().mi_policy_bwd_bf16
    extra = ()
    if ws:
        extra = (None, None)
        if masks is not None:
            am, cm = masks
            _need(len(am) >= La - 1 and len(cm) >= Lc - 1 and all(
                t is not None and t.dtype == torch.uint8 for t in [*am[:La - 1], *cm[:Lc - 1]]),
                "policy_bwd_bf16: one uint8 mask per hidden layer")
            extra = (arr(am[:La - 1], La - 1), arr(cm[:Lc - 1], Lc - 1))
    check(entry(
        ptr(mean_and_std, f32), ptr(extras, f32), ptr(rng_state), int(offset_add),
        ptr(eps2, f32), ptr(g_ll, f32), float(g_reg), float(min_std), float(std_scale),
        float(entropy_weight), ptr(g_value, f32), M,
        La, arr(a_w, La), i64s(a_dims), i64s(a_acts), arr(a_aux[:La - 1], La - 1),
        ptr(a_dz[La - 1], bf16), arr(a_dz[:La - 1], La - 1),
        Lc, arr(c_w, Lc), i64s(c_dims), i64s(c_acts), arr(c_aux[:Lc - 1], Lc - 1),
        ptr(c_dz[Lc - 1], bf16), arr(c_dz[:Lc - 1], Lc - 1), *extra, stream()),
        "mi_policy_ws_bwd_bf16" if ws else "mi_policy_bwd_bf16")
    return a_dz, c_dz


def policy_bwd_gae_supported(T: int, B: int, actor, critic) -> bool:
    """`mi_policy_ws_bwd_gae_supported` for trunks described as in `policy_bwd_bf16`."""
    (_, a_dims, a_acts, _), (_, c_dims, c_acts, _) = actor, critic
    i64s = lambda v: (ctypes.c_int64 * len(v))(*[int(q) for q in v])
    return bool(lib().mi_policy_ws_bwd_gae_supported(
        int(T), int(B), len(a_acts), i64s(a_dims), i64s(a_acts), len(c_acts), i64s(c_dims),
        i64s(c_acts)))


def policy_bwd_gae_bf16(mean_and_std, extras, rng_state, offset_add: int, g_reg: float, actor,
                        critic, masks, rewards, values, last_value, done, truncated, ll_new,
                        ll_old, reg, gamma: float, lambda_: float, normalize: bool,
                        clip_range: float, critic_weight: float, *, min_std: float,
                        std_scale: float, entropy_weight: float, eps2=None,
                        loss_out: torch.Tensor | None = None, defer: list | None = None,
                        comm=None):
    """`comm` (a `comm.PeerComm`): env-sharded run — the advantage statistics are exchanged
    between the ranks INSIDE the launch (global-minibatch normalisation, `ppo.py:477-480`).

    `gae_ppo_loss` + `policy_bwd_bf16(ws=True, masks=...)` in ONE launch
    (`mi_policy_ws_bwd_gae_bf16`): the GAE scan, the advantage statistics and the loss
    gradients are evaluated inside the backward's workgroups.  `[T, B]` operands as in
    `gae_ppo_loss`.  Returns (actor dz list, critic dz list, loss_out[4]); the dz images are
    bit-identical to the two launches, the four scalars equal up to fp64 summation order.
    `defer` (a list): the launch leaves its per-tile partials in a buffer of their own and
    appends (partials, rows, T * B, loss_out) to the list instead of summing them at its tail;
    `policy_loss_finalize(defer)` fills every pending `loss_out` in one launch (same bits)."""
    T, B = rewards.shape
    M, A2 = mean_and_std.shape
    dev = mean_and_std.device
    _need(M == T * B, "policy_bwd_gae_bf16: mean_and_std must have T * B rows")
    for t in (values, ll_new, ll_old, done, truncated):
        _need(t.shape == (T, B) and t.is_contiguous(),
              "policy_bwd_gae_bf16: operands must be contiguous [T, B]")
    _need(rewards.is_contiguous() and last_value.shape == (B,) and last_value.is_contiguous(),
          "policy_bwd_gae_bf16: rewards [T, B], last_value [B], contiguous")
    _need(reg is None or (reg.numel() == M and reg.is_contiguous()),
          "policy_bwd_gae_bf16: reg must be [T, B]")
    (a_w, a_dims, a_acts, a_aux), (c_w, c_dims, c_acts, c_aux) = actor, critic
    La, Lc = len(a_w), len(c_w)
    _need(extras.shape == (M, A2 // 2), "policy_bwd_gae_bf16: shapes")
    am, cm = masks
    _need(len(am) >= La - 1 and len(cm) >= Lc - 1 and all(
        t is not None and t.dtype == torch.uint8 for t in [*am[:La - 1], *cm[:Lc - 1]]),
        "policy_bwd_gae_bf16: one uint8 mask per hidden layer")
    a_dz = [_bf_buf(M, a_dims[l + 1], dev) for l in range(La)]
    c_dz = [_bf_buf(M, c_dims[l + 1], dev) for l in range(Lc)]
    if loss_out is None:
        loss_out = torch.empty(4, dtype=f32, device=dev)
    arr = lambda ts, n: (ctypes.c_void_p * max(n, 1))(*[ptr(t) for t in ts])
    i64s = lambda v: (ctypes.c_int64 * len(v))(*[int(q) for q in v])
    # (before the work annotation: the size query is a recorded C-ABI call of its own)
    ws = workspace(dev, "policy_bwd_gae", lib().mi_policy_ws_bwd_gae_workspace_bytes(M),
                   zeroed=True)
    part = None
    if defer is not None:
        part = torch.empty(M // 64 * 4, dtype=f64, device=dev)
        defer.append((part, M // 64, M, loss_out))
    if profiler.active:
        flop = sum(a_dims[l] * a_dims[l + 1] for l in range(1, La)) \
            + sum(c_dims[l] * c_dims[l + 1] for l in range(1, Lc))
        profiler.next_flops = 2.0 * M * flop
        moved = sum(t.numel() * t.element_size()
                    for t in [*am[:La - 1], *cm[:Lc - 1], *a_dz, *c_dz])
        w_bytes = sum(2 * a_dims[l] * a_dims[l + 1] for l in range(1, La)) \
            + sum(2 * c_dims[l] * c_dims[l + 1] for l in range(1, Lc))
        # sampler rows (mean_and_std, raw action) + the GAE / loss operands, each once
        # algorithmically: rewards, values, log-likelihoods (x2), regulariser, two flag bytes
        profiler.next_bytes = (4.0 * M * (A2 + A2 // 2) + M * (4.0 * 5 + 2) + w_bytes + moved)
    check(lib().mi_policy_ws_bwd_gae_bf16(
        ptr(mean_and_std, f32), ptr(extras, f32), ptr(rng_state), int(offset_add),
        ptr(eps2, f32), float(g_reg), float(min_std), float(std_scale), float(entropy_weight),
        ptr(rewards, f32), ptr(values, f32), ptr(last_value, f32), ptr(_as_u8(done), u8),
        ptr(_as_u8(truncated), u8), ptr(ll_new, f32), ptr(ll_old, f32), ptr(reg, f32),
        float(gamma), float(lambda_), int(bool(normalize)), float(clip_range),
        float(critic_weight), None if part is not None else ptr(loss_out, f32), ptr(part, f64),
        ptr(ws), T, B,
        La, arr(a_w, La), i64s(a_dims), i64s(a_acts), arr(a_aux[:La - 1], La - 1),
        ptr(a_dz[La - 1], bf16), arr(a_dz[:La - 1], La - 1),
        Lc, arr(c_w, Lc), i64s(c_dims), i64s(c_acts), arr(c_aux[:Lc - 1], Lc - 1),
        ptr(c_dz[Lc - 1], bf16), arr(c_dz[:Lc - 1], Lc - 1),
        arr(am[:La - 1], La - 1), arr(cm[:Lc - 1], Lc - 1),
        None if comm is None else comm._h, stream()),
        "mi_policy_ws_bwd_gae_bf16")
    return a_dz, c_dz, loss_out


def policy_loss_finalize(pending: list) -> None:
    """Sum the per-tile loss partials of deferred `policy_bwd_gae_bf16` launches
    or `gae_ppo_loss` launches (`mi_policy_loss_finalize_f32`): `pending` = [(partials, rows of
    partials, T * B, loss_out[4])], cleared."""
    for i in range(0, len(pending), 32):
        chunk = pending[i:i + 32]
        n = len(chunk)
        P = ctypes.c_void_p * n
        I = ctypes.c_int64 * n
        check(lib().mi_policy_loss_finalize_f32(
            n, P(*[ptr(c[0], f64) for c in chunk]), I(*[int(c[1]) for c in chunk]),
            I(*[int(c[2]) for c in chunk]), P(*[ptr(c[3], f32) for c in chunk]), stream()),
            "mi_policy_loss_finalize_f32")
    pending.clear()


# workspaces whose header carries a sticky count of timed-out in-kernel hand-overs (header
# word 2) and the test hook (words 3, 4): csrc/trunk_ws.hip policy_ws_bwd_gae_kernel,
# csrc/gae_loss.hip gae_loss_kernel
_HANDOVER_TAGS = ("policy_bwd_gae", "gae_loss")


def health_words(device=None) -> list:
    """One 1-element int32 view per sticky timeout word that exists on `device` (all devices
    if None): the advantage-statistics hand-overs of `mi_policy_ws_bwd_gae_bf16` and
    `mi_gae_ppo_loss_f32`, and the one-shot peer exchange's error word.  The training loop
    copies them to the host with every iteration's metrics (`loop.MetricPack`) and raises in
    the iteration a word turns non-zero."""
    out = []
    for key, ws in _workspaces.items():
        if key[2] in _HANDOVER_TAGS and (device is None or key[0] == str(device)):
            out.append(ws[8:12].view(torch.int32))
    from . import parallel

    comm = parallel.peer_comm()
    if comm is not None and (device is None or str(comm.device) == str(device)):
        out.append(comm.error_word)
    return out


def handover_timeouts() -> int:
    """How many bounded in-kernel waits of the advantage-statistics hand-overs
    (`mi_policy_ws_bwd_gae_bf16`, `mi_gae_ppo_loss_f32`) have run out since the workspaces
    were created.  Reads device words (it synchronises).  Non-zero = some gradient step saw
    incomplete statistics (the kernels poison that step's normalisation with NaN)."""
    total = 0
    for key, ws in _workspaces.items():
        if key[2] in _HANDOVER_TAGS:
            total += int(ws[8:12].view(torch.int32).item())
    return total


policy_bwd_gae_timeouts = handover_timeouts  # round-2 name


def set_handover_test_hook(limit_us: int = 0, extra_arrivals: int = 0) -> int:
    """TEST HOOK.  Make the hand-over waits of every existing workspace give up after
    `limit_us` microseconds and wait for `extra_arrivals` arrivals that never come (0, 0
    restores production behaviour: 2 s, none).  Returns the number of workspaces touched."""
    n = 0
    for key, ws in _workspaces.items():
        if key[2] in _HANDOVER_TAGS:
            ws[12:20].view(torch.int32).copy_(
                torch.tensor([int(limit_us), int(extra_arrivals)], dtype=torch.int32))
            n += 1
    return n


def clear_handover_timeouts() -> None:
    """TEST HOOK: zero the sticky words (after a test that forced a timeout)."""
    for key, ws in _workspaces.items():
        if key[2] in _HANDOVER_TAGS:
            ws[8:12].zero_()


def mlp_ws_bwd_dx_bf16(g_out: torch.Tensor, w_bfs: list, dims: list, acts: list, auxs: list):
    """`mlp_bwd_dx_bf16(..., need_input_grad=False)` with a linear last layer on the
    weights-stationary kernel.  Returns the dz list."""
    M = g_out.shape[0]
    L = len(w_bfs)
    _need(L >= 2 and len(dims) == L + 1 and g_out.shape[1] == dims[-1], "mlp_ws_bwd_dx_bf16: dims")
    dev = g_out.device
    dz = [_bf_buf(M, dims[l + 1], dev) for l in range(L)]
    P = ctypes.c_void_p * L
    Pm = ctypes.c_void_p * (L - 1)
    check(lib().mi_mlp_ws_bwd_dx_bf16(
        ptr(g_out, f32), M, L, P(*[ptr(w, bf16) for w in w_bfs]),
        (ctypes.c_int64 * (L + 1))(*[int(d) for d in dims]),
        (ctypes.c_int64 * L)(*[int(a) for a in acts]), Pm(*[ptr(a) for a in auxs[:L - 1]]),
        ptr(dz[L - 1], bf16), Pm(*[ptr(t, bf16) for t in dz[:L - 1]]), stream()),
        "mi_mlp_ws_bwd_dx_bf16")
    return dz


def mlp_bwd_dx_bf16(g_out: torch.Tensor, aux_last, act_last: int, w_bfs: list, dims: list,
                    acts: list, auxs: list, need_input_grad: bool):
    """Fused dX chain of an MLP trunk.  Returns (dz list per layer [L], g_in | None):
    dz[l] is the bf16 gradient w.r.t. layer l's pre-activation, [M, pad8(N_l)]."""
    M = g_out.shape[0]
    L = len(w_bfs)
    _need(len(dims) == L + 1 and g_out.shape[1] == dims[-1], "mlp_bwd_dx_bf16: dims")
    dev = g_out.device
    dz = [_bf_buf(M, dims[l + 1], dev) for l in range(L)]
    g_in = torch.empty(M, dims[0], dtype=f32, device=dev) if need_input_grad else None
    P = ctypes.c_void_p * L
    Pm = ctypes.c_void_p * max(L - 1, 1)
    if profiler.active:
        first = 0 if need_input_grad else 1
        profiler.next_flops = 2.0 * M * sum(dims[l] * dims[l + 1] for l in range(first, L))
        w_bytes = sum(2 * dims[l] * dims[l + 1] for l in range(first, L))
        moved = sum(t.numel() * 2 for t in [aux_last, *auxs[:L - 1], *dz] if t is not None)
        # fp32 output gradient + act' operands + weights in, every dz (+ fp32 g_in) out
        profiler.next_bytes = (4.0 * M * dims[-1] + w_bytes + moved
                               + (4.0 * M * dims[0] if need_input_grad else 0.0))
    check(lib().mi_mlp_bwd_dx_bf16(
        ptr(g_out, f32), ptr(aux_last), int(act_last), M, L, P(*[ptr(w, bf16) for w in w_bfs]),
        (ctypes.c_int64 * (L + 1))(*[int(d) for d in dims]),
        (ctypes.c_int64 * L)(*[int(a) for a in acts]),
        Pm(*[ptr(a) for a in auxs[:L - 1]]) if L > 1 else None, ptr(dz[L - 1], bf16),
        Pm(*[ptr(t, bf16) for t in dz[:L - 1]]) if L > 1 else None, ptr(g_in, f32), stream()),
        "mi_mlp_bwd_dx_bf16")
    return dz, g_in


# ------------------------------------------------------------- a14: loss
def _loss_ws(device):
    return workspace(device, "loss", lib().mi_ppo_loss_workspace_bytes(1), zeroed=True)


def adv_stats(adv: torch.Tensor) -> torch.Tensor:
    """fp64 [3] = (sum, sum of squares, count) of the advantages."""
    n = adv.numel()
    stats = torch.empty(3, dtype=f64, device=adv.device)
    check(lib().mi_adv_stats_f32(ptr(adv, f32), n, ptr(stats, f64), ptr(_loss_ws(adv.device)),
                                 stream()), "mi_adv_stats_f32")
    return stats


def ppo_loss(ll_new, ll_old, adv, values, reg, stats, clip_range: float, critic_weight: float,
             loss_out: torch.Tensor | None = None):
    """Returns (g_ll, g_v, loss_out[4] = actor, critic, regularization, clip_frac)."""
    n = adv.numel()
    _need((ll_new is None) == (ll_old is None), "ppo_loss: ll_new and ll_old go together")
    _need(ll_new is not None or values is not None, "ppo_loss: nothing to evaluate")
    for t in (ll_new, ll_old, values, reg):
        _need(t is None or t.numel() == n, "ppo_loss: operand sizes differ")
    dev = adv.device
    g_ll = torch.empty(n, dtype=f32, device=dev) if ll_new is not None else None
    g_v = torch.empty(n, dtype=f32, device=dev) if values is not None else None
    if loss_out is None:
        loss_out = torch.empty(4, dtype=f32, device=dev)
    check(lib().mi_ppo_loss_f32(ptr(ll_new, f32), ptr(ll_old, f32), ptr(adv, f32),
                                ptr(values, f32), ptr(reg, f32), ptr(stats, f64),
                                float(clip_range), float(critic_weight), ptr(g_ll, f32),
                                ptr(g_v, f32), ptr(loss_out, f32), ptr(_loss_ws(dev)), n,
                                stream()), "mi_ppo_loss_f32")
    return g_ll, g_v, loss_out


def gae_ppo_loss_supported(T: int, N: int) -> bool:
    return bool(lib().mi_gae_ppo_loss_supported(int(T), int(N)))


def gae_ppo_loss(rewards, values, last_value, done, truncated, ll_new, ll_old, reg, gamma: float,
                 lambda_: float, normalize: bool, clip_range: float, critic_weight: float,
                 loss_out: torch.Tensor | None = None, want_adv: bool = False,
                 defer: list | None = None):
    """GAE + advantage statistics + loss terms and gradients in one launch
    (`mi_gae_ppo_loss_f32`).  All operands `[T, N]` (last_value `[N]`), reg may be None.
    Returns (g_ll [T,N], g_v [T,N], loss_out [4], advantages | None).  `defer` (a list): the
    four scalars are left as per-workgroup partials and `loss_out` is filled by
    `policy_loss_finalize(defer)` (same bits)."""
    T, N = rewards.shape
    for t in (values, ll_new, ll_old, done, truncated):
        _need(t.shape == (T, N), "gae_ppo_loss: operands must be [T, N]")
    _need(reg is None or reg.numel() == T * N, "gae_ppo_loss: reg must be [T, N]")
    _need(last_value.shape == (N,), "gae_ppo_loss: last_value must be [N]")
    dev = rewards.device
    g_ll = torch.empty(T, N, dtype=f32, device=dev)
    g_v = torch.empty(T, N, dtype=f32, device=dev)
    adv = torch.empty(T, N, dtype=f32, device=dev) if want_adv else None
    if loss_out is None:
        loss_out = torch.empty(4, dtype=f32, device=dev)
    ws = workspace(dev, "gae_loss", lib().mi_gae_ppo_loss_workspace_bytes(), zeroed=True)
    part = None
    if defer is not None:
        rows = (N + 63) // 64
        part = torch.empty(rows * 4, dtype=f64, device=dev)
        defer.append((part, rows, T * N, loss_out))
    check(lib().mi_gae_ppo_loss_f32(
        ptr(rewards, f32), ptr(values, f32), ptr(last_value, f32), ptr(_as_u8(done), u8),
        ptr(_as_u8(truncated), u8), ptr(ll_new, f32), ptr(ll_old, f32), ptr(reg, f32),
        float(gamma), float(lambda_), int(bool(normalize)), float(clip_range),
        float(critic_weight), ptr(adv, f32), None, ptr(g_ll, f32), ptr(g_v, f32),
        None if part is not None else ptr(loss_out, f32), ptr(part, f64), ptr(ws), T, N,
        stream()), "mi_gae_ppo_loss_f32")
    return g_ll, g_v, loss_out, adv


# -------------------------------------------------------- a15: optimiser
def begin_grad_step(grads: torch.Tensor, step: torch.Tensor | None) -> None:
    check(lib().mi_begin_grad_step_f32(ptr(grads, f32), grads.numel(), ptr(step, i64), stream()),
          "mi_begin_grad_step_f32")


def global_norm(grads: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    n = grads.numel()
    out = out if out is not None else torch.empty(1, dtype=f32, device=grads.device)
    ws = workspace(grads.device, "gnorm", lib().mi_global_norm_workspace_bytes(n))
    check(lib().mi_global_norm_f32(ptr(grads, f32), n, ptr(out, f32), ptr(ws), stream()),
          "mi_global_norm_f32")
    return out


def adam_step(params, grads, m, v, step, *, lr: float, b1: float = 0.9, b2: float = 0.999,
              eps: float = 1e-8, weight_decay: float = 0.0, grad_norm=None,
              max_norm: float = 0.0, begin_next: bool = False, shadows=None,
              slabs=None) -> None:
    """`begin_next`: `step` counts completed steps; this launch advances it and leaves
    `grads` zeroed (it doubles as the next step's `begin_grad_step`).  `shadows`: up to
    16 tuples (begin, K, N, w_bf, wt_bf, frag_fwd, frag_bwd) of Dense kernels stored in
    `params` whose bf16 images are written by the same launch.  `slabs`: pending dW slabs
    (slab_ptrs, S, Ks, Ns, gw_offsets, gb_offsets[, bias_first]) summed into the gradient as it
    is read (`mi_adam_step_slabs_f32`)."""
    n = params.numel()
    for t in (grads, m, v):
        _need(t.numel() == n, "adam_step: arena sizes differ")
    ticket = workspace(params.device, "adam_ticket", 16, zeroed=True) if begin_next else None
    sh = list(shadows or [])[:16]
    ns = len(sh)
    I = ctypes.c_int64 * max(ns, 1)
    P = ctypes.c_void_p * max(ns, 1)
    col = lambda j: [t[j] for t in sh]
    args = (ptr(params, f32), ptr(grads, f32), ptr(m, f32), ptr(v, f32), n,
            float(lr), float(b1), float(b2), float(eps),
            float(weight_decay), ptr(step, i64), ptr(grad_norm, f32),
            float(max_norm), ptr(ticket), ns,
            I(*col(0)) if ns else None, I(*col(1)) if ns else None,
            I(*col(2)) if ns else None,
            P(*[ptr(t, bf16) for t in col(3)]) if ns else None,
            P(*[ptr(t, bf16) for t in col(4)]) if ns else None,
            P(*[ptr(t, bf16) for t in col(5)]) if ns else None,
            P(*[ptr(t, bf16) for t in col(6)]) if ns else None)
    if slabs is None:
        check(lib().mi_adam_step_f32(*args, stream()), "mi_adam_step_f32")
        return
    _need(begin_next, "adam_step: pending slabs need begin_next (the launch consumes them)")
    sp, S, Ks, Ns, gw_off, gb_off = slabs[:6]
    b_lo = slabs[6] if len(slabs) > 6 else None
    nl = len(Ks)
    Pl = ctypes.c_void_p * nl
    Il = ctypes.c_int64 * nl
    check(lib().mi_adam_step_slabs_f32(*args, nl, Pl(*sp), Il(*S), Il(*Ks), Il(*Ns),
                                       Il(*gw_off), Il(*gb_off),
                                       Il(*b_lo) if b_lo is not None and any(b_lo) else None,
                                       stream()),
          "mi_adam_step_slabs_f32")


# ------------------------------------------------------ a5 / a7: movement
def gather_cols(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """`src[:, idx]` for a time-major `[T, N, ...]` tensor of any dtype."""
    _need(src.dim() >= 2, "gather_cols: src must be [T, N, ...]")
    _need(idx.dim() == 1 and idx.dtype == i64, "gather_cols: idx must be int64 [L]")
    T, N = src.shape[:2]
    L = idx.numel()
    row_bytes = src.element_size()
    for s in src.shape[2:]:
        row_bytes *= s
    dst = torch.empty((T, L, *src.shape[2:]), dtype=src.dtype, device=src.device)
    if row_bytes == 0:
        return dst
    s_ = src.view(torch.uint8) if src.dtype == torch.bool else src
    d_ = dst.view(torch.uint8) if dst.dtype == torch.bool else dst
    check(lib().mi_gather_cols(ptr(s_), ptr(idx, i64), ptr(d_), T, N, L, row_bytes, stream()),
          "mi_gather_cols")
    return dst


def gather_cols_multi(leaves: list, idx: torch.Tensor, groups: int = 1) -> list:
    """`gather_cols` for several time-major `[T_l, N, ...]` leaves with ONE launch per
    16 leaves (all leaves share N and the index vector).  `groups` > 1: `idx` is that
    many consecutive index groups (minibatches) and every output is
    `[groups, T_l, L / groups, ...]` — each group a contiguous time-major block."""
    _need(idx.dim() == 1 and idx.dtype == i64, "gather_cols_multi: idx must be int64 [L]")
    L = idx.numel()
    _need(groups >= 1 and L % groups == 0, "gather_cols_multi: groups must divide len(idx)")
    GL = L // groups if L else 1
    outs: list = [None] * len(leaves)
    batch: list = []
    N = None

    def flush():
        if not batch:
            return
        n = len(batch)
        P = ctypes.c_void_p * n
        I = ctypes.c_int64 * n
        check(lib().mi_gather_cols_multi(P(*[b[0] for b in batch]), P(*[b[1] for b in batch]),
                                         I(*[b[2] for b in batch]), I(*[b[3] for b in batch]), n,
                                         ptr(idx, i64), N, L, GL, stream()),
              "mi_gather_cols_multi")
        batch.clear()

    for k, src in enumerate(leaves):
        _need(src.dim() >= 2, "gather_cols_multi: leaves must be [T, N, ...]")
        if N is None:
            N = src.shape[1]
        _need(src.shape[1] == N, "gather_cols_multi: leaves must share N")
        row_bytes = src.element_size()
        for d in src.shape[2:]:
            row_bytes *= d
        shape = ((src.shape[0], L, *src.shape[2:]) if groups == 1
                 else (groups, src.shape[0], GL, *src.shape[2:]))
        dst = torch.empty(shape, dtype=src.dtype, device=src.device)
        outs[k] = dst
        if row_bytes == 0 or L == 0 or src.shape[0] == 0:
            continue
        s_ = src.view(torch.uint8) if src.dtype == torch.bool else src
        d_ = dst.view(torch.uint8) if dst.dtype == torch.bool else dst
        batch.append((ptr(s_), ptr(d_), src.shape[0], row_bytes))
        if len(batch) == 16:
            flush()
    flush()
    return outs


def select_rows(mask: torch.Tensor, on_true: torch.Tensor, on_false: torch.Tensor,
                out: torch.Tensor | None = None) -> torch.Tensor:
    """`where(mask[:, None...], on_true, on_false)` over the leading axis; `on_true`
    may be a single row (shape `on_false.shape[1:]`) that is broadcast."""
    B = mask.shape[0]
    _need(on_false.shape[0] == B, "select_rows: leading dim must equal the mask's")
    row_bytes = on_false.element_size()
    for s in on_false.shape[1:]:
        row_bytes *= s
    if on_true.shape == on_false.shape:
        stride = row_bytes
    elif on_true.shape == on_false.shape[1:]:
        stride = 0
    else:
        raise MippoError("select_rows: on_true must match on_false or be one row")
    _need(on_true.dtype == on_false.dtype, "select_rows: dtype mismatch")
    out = out if out is not None else torch.empty_like(on_false)
    if row_bytes == 0 or B == 0:
        return out
    v = lambda t: t.view(torch.uint8) if t.dtype == torch.bool else t
    check(lib().mi_select_rows(ptr(_as_u8(mask)), ptr(v(on_true)), stride, ptr(v(on_false)),
                               ptr(v(out)), B, row_bytes, stream()), "mi_select_rows")
    return out


def select_rows_multi(mask: torch.Tensor, pairs: list) -> list:
    """`select_rows` for several (on_true, on_false) leaves in ONE launch per 16
    leaves.  Returns the list of outputs in order."""
    B = mask.shape[0]
    outs: list = [None] * len(pairs)
    batch: list = []
    v = lambda t: t.view(torch.uint8) if t.dtype == torch.bool else t

    def flush():
        if not batch:
            return
        n = len(batch)
        P = ctypes.c_void_p * n
        L = ctypes.c_int64 * n
        check(lib().mi_select_rows_multi(
            ptr(_as_u8(mask)), P(*[b[1] for b in batch]), L(*[b[2] for b in batch]),
            P(*[b[3] for b in batch]), P(*[b[4] for b in batch]), L(*[b[5] for b in batch]), n, B,
            stream()), "mi_select_rows_multi")
        batch.clear()

    for k, (on_true, on_false) in enumerate(pairs):
        _need(on_false.shape[0] == B, "select_rows_multi: leading dim must equal the mask's")
        _need(on_true.dtype == on_false.dtype, "select_rows_multi: dtype mismatch")
        row_bytes = on_false.element_size()
        for d in on_false.shape[1:]:
            row_bytes *= d
        if on_true.shape == on_false.shape:
            stride = row_bytes
        elif on_true.shape == on_false.shape[1:]:
            stride = 0
        else:
            raise MippoError("select_rows_multi: on_true must match on_false or be one row")
        out = torch.empty_like(on_false)
        outs[k] = out
        if row_bytes == 0 or B == 0:
            continue
        batch.append((k, ptr(v(on_true)), stride, ptr(v(on_false)), ptr(v(out)), row_bytes))
        if len(batch) == 16:
            flush()
    flush()
    return outs


# ------------------------------------------------- a4 / a18: keys, episodes
KEY_SPLIT, KEY_BITS, KEY_RANDINT, KEY_UNIFORM, KEY_UNIT_UNIFORM = 0, 1, 2, 3, 4


def copy_multi(pairs: list) -> None:
    """`dst.copy_(src)` for a list of (dst, src) contiguous same-shape tensors, 16 per
    launch."""
    for i in range(0, len(pairs), 16):
        grp = pairs[i:i + 16]
        n = len(grp)
        for d, s in grp:
            _need(d.shape == s.shape and d.dtype == s.dtype and d.is_contiguous()
                  and s.is_contiguous(), "copy_multi: pairs must match and be contiguous")
        P = ctypes.c_void_p * n
        check(lib().mi_copy_multi(P(*[ptr(s) for _, s in grp]), P(*[ptr(d) for d, _ in grp]),
                                  (ctypes.c_int64 * n)(*[d.numel() * d.element_size()
                                                         for d, _ in grp]), n, stream()),
              "mi_copy_multi")


def stack_multi(groups: list) -> list:
    """`torch.stack(group, 0)` for every group of a list of equal-length groups of
    same-shape contiguous tensors — one launch per 16 leaves / 448 segments
    (`mi_stack_multi`) instead of one per group."""
    if not groups:
        return []
    T = len(groups[0])
    outs = []
    per = max(1, min(16, 448 // max(T, 1)))
    for g in groups:
        _need(len(g) == T, "stack_multi: groups must have one length")
        x0 = g[0]
        _need(all(x.shape == x0.shape and x.dtype == x0.dtype and x.is_contiguous() for x in g),
              "stack_multi: a group's tensors must share shape / dtype and be contiguous")
        outs.append(torch.empty((T, *x0.shape), dtype=x0.dtype, device=x0.device))
    if T == 0 or T > 448:
        return [torch.stack(g, 0) if g else o for g, o in zip(groups, outs)]
    for i in range(0, len(groups), per):
        grp, dst = groups[i:i + per], outs[i:i + per]
        n = len(grp)
        P = ctypes.c_void_p * (n * T)
        check(lib().mi_stack_multi(
            P(*[ptr(x) for g in grp for x in g]), (ctypes.c_void_p * n)(*[ptr(d) for d in dst]),
            (ctypes.c_int64 * n)(*[g[0].numel() * g[0].element_size() for g in grp]), n, T,
            stream()), "mi_stack_multi")
    return outs


def key_expand(keys: torch.Tensor, m: int, mode: int, minval: int = 0, maxval: int = 0,
               child_major: bool = False, fold: Optional[torch.Tensor] = None):
    """keys (int64, any shape) -> `[*keys.shape, m]` children / bits / integers / floats
    (`[m, *keys.shape]` when child_major: each child set is contiguous).  `fold`
    (int64, keys' shape) is folded into the keys first, as `key_fold` would."""
    _need(keys.dtype == i64, "key_expand: keys must be int64")
    k = keys if keys.is_contiguous() else keys.contiguous()
    n = k.numel()
    if fold is not None:
        _need(fold.dtype == i64 and fold.shape == k.shape, "key_expand: fold must match keys")
        fold = fold if fold.is_contiguous() else fold.contiguous()
    dt = f32 if mode in (KEY_UNIFORM, KEY_UNIT_UNIFORM) else i64
    shape = (m, *k.shape) if child_major else (*k.shape, m)
    out = torch.empty(shape, dtype=dt, device=k.device)
    check(lib().mi_key_expand(ptr(k, i64), ptr(fold, i64), ptr(out), n, int(m), int(mode),
                              int(minval), int(maxval), int(bool(child_major)), stream()),
          "mi_key_expand")
    return out


def mock_env_step(key: torch.Tensor, step_count: torch.Tensor, max_steps: int, widths: list):
    """MockEnv.step in one launch: returns (step_count + 1, done (bool), [obs leaf [n, w]...])."""
    _need(key.dtype == i64 and step_count.dtype == i64 and key.shape == step_count.shape
          and key.dim() == 1, "mock_env_step: key / step_count must be int64 [n]")
    _need(1 <= len(widths) <= 8, "mock_env_step: 1..8 observation leaves")
    n = key.shape[0]
    dev = key.device
    count_out = torch.empty_like(step_count)
    done = torch.empty(n, dtype=torch.bool, device=dev)
    leaves = [torch.empty(n, int(w), dtype=f32, device=dev) for w in widths]
    nl = len(widths)
    check(lib().mi_mock_env_step(
        ptr(key.contiguous(), i64), ptr(step_count.contiguous(), i64), int(max_steps),
        ptr(count_out, i64), ptr(done.view(u8)), (ctypes.c_void_p * nl)(*[ptr(t) for t in leaves]),
        (ctypes.c_int64 * nl)(*[int(w) for w in widths]), nl, n, stream()), "mi_mock_env_step")
    return count_out, done, leaves


def rollout_mock_ws_supported(a_dims, a_acts, c_dims, c_acts) -> bool:
    i64s = lambda v: (ctypes.c_int64 * len(v))(*[int(q) for q in v])
    return bool(lib().mi_rollout_mock_ws_supported(len(a_acts), i64s(a_dims), i64s(a_acts),
                                                   len(c_acts), i64s(c_dims), i64s(c_acts)))


def rollout_mock_ws(env_key, env_step_count, wrap_step_counter, obs0, reset_key,
                    max_steps: int, max_len: int, T: int, norm, actor, critic, rng_state,
                    offset_add: int, *, min_std: float, std_scale: float,
                    entropy_weight: float, deterministic: bool) -> dict:
    """The whole T-step rollout of EpisodeWrapper(MockEnv) under the fused MLP actor-critic in
    ONE launch (`mi_rollout_mock_ws_bf16`).  `norm` / `actor` / `critic` as `policy_fwd_bf16`.
    Returns the Transition leaves (time-major) and the state after the last reset select."""
    N, K0 = obs0.shape
    dev = obs0.device
    (a_w, a_b, a_dims, a_acts), (c_w, c_b, c_dims, c_acts) = actor, critic
    La, Lc = len(a_w), len(c_w)
    A = a_dims[-1] // 2
    for t in (env_key, env_step_count, wrap_step_counter):
        _need(t.dtype == i64 and t.shape == (N,) and t.is_contiguous(),
              "rollout_mock_ws: env state leaves must be contiguous int64 [N]")
    _need(reset_key.dtype == i64 and reset_key.numel() == 1, "rollout_mock_ws: one int64 key")
    _need(obs0.is_contiguous() and a_dims[0] == K0 and c_dims[0] == K0,
          "rollout_mock_ws: obs must be a contiguous [N, K0] matching the trunks")
    mk = lambda *sh: torch.empty(*sh, dtype=f32, device=dev)
    out = dict(
        obs=mk(T, N, K0), next_obs=mk(T, N, K0), reward=mk(T, N),
        done=torch.empty(T, N, dtype=torch.bool, device=dev),
        truncated=torch.empty(T, N, dtype=torch.bool, device=dev),
        raw=mk(T, N, A), action=mk(T, N, A), log_likelihood=mk(T, N), mu=mk(T, N, A),
        sigma=mk(T, N, A), value=mk(T, N, c_dims[-1]),
        key_out=torch.empty_like(env_key), count_out=torch.empty_like(env_step_count),
        counter_out=torch.empty_like(wrap_step_counter), obs_out=mk(N, K0), reward_out=mk(N))
    arr = lambda ts, L: (ctypes.c_void_p * L)(*[ptr(t) for t in ts])
    i64s = lambda v: (ctypes.c_int64 * len(v))(*[int(q) for q in v])
    n_mean, n_m2, n_cnt, n_eps = norm if norm is not None else (None, None, None, 0.0)
    if profiler.active:
        flop = sum(a_dims[l] * a_dims[l + 1] for l in range(La)) \
            + sum(c_dims[l] * c_dims[l + 1] for l in range(Lc))
        profiler.next_flops = 2.0 * T * N * flop
        profiler.next_bytes = 4.0 * T * N * (2 * K0 + 4 * A + 2 + c_dims[-1]) + 2.0 * T * N
    check(lib().mi_rollout_mock_ws_bf16(
        ptr(env_key, i64), ptr(env_step_count, i64), ptr(wrap_step_counter, i64), ptr(obs0, f32),
        ptr(reset_key.reshape(1), i64), int(max_steps), int(max_len), int(T), N,
        ptr(n_mean, f32), ptr(n_m2, f32), ptr(n_cnt, f32), float(n_eps),
        La, arr(a_w, La), arr(a_b, La), i64s(a_dims), i64s(a_acts),
        Lc, arr(c_w, Lc), arr(c_b, Lc), i64s(c_dims), i64s(c_acts),
        ptr(rng_state), int(offset_add), float(min_std), float(std_scale), float(entropy_weight),
        int(bool(deterministic)), ptr(out["obs"], f32), ptr(out["next_obs"], f32),
        ptr(out["reward"], f32), ptr(out["done"].view(u8)), ptr(out["truncated"].view(u8)),
        ptr(out["raw"], f32), ptr(out["action"], f32), ptr(out["log_likelihood"], f32),
        ptr(out["mu"], f32), ptr(out["sigma"], f32), ptr(out["value"], f32),
        ptr(out["key_out"], i64), ptr(out["count_out"], i64), ptr(out["counter_out"], i64),
        ptr(out["obs_out"], f32), ptr(out["reward_out"], f32), stream()),
        "mi_rollout_mock_ws_bf16")
    return out


def rollout_mock_gru_ws(env_key, env_step_count, wrap_step_counter, obs0, reset_key,
                        max_steps: int, max_len: int, T: int, norm, dense_in, proj, w_h, b_hn,
                        dense_out, h_in, critic, rng_state, offset_add: int, *, min_std: float,
                        std_scale: float, entropy_weight: float, deterministic: bool) -> dict:
    """`rollout_mock_ws` for the recurrent actor-critic of `gru_policy_step`
    (`mi_rollout_mock_gru_ws_bf16`): the carry rides in the launch; `h_out` is the carry after
    the last reset select."""
    N, K0 = obs0.shape
    H = h_in.shape[1]
    dev = obs0.device
    c_w, c_b, c_dims, c_acts = critic
    Lc = len(c_w)
    A2 = dense_out[1].shape[0]
    A = A2 // 2
    for t in (env_key, env_step_count, wrap_step_counter):
        _need(t.dtype == i64 and t.shape == (N,) and t.is_contiguous(),
              "rollout_mock_gru_ws: env state leaves must be contiguous int64 [N]")
    _need(reset_key.dtype == i64 and reset_key.numel() == 1, "rollout_mock_gru_ws: one int64 key")
    _need(h_in.shape == (N, H) and h_in.is_contiguous() and w_h.shape == (H, 3 * H)
          and obs0.is_contiguous(), "rollout_mock_gru_ws: carry [N, H], kernel [H, 3H]")
    mk = lambda *sh: torch.empty(*sh, dtype=f32, device=dev)
    out = dict(
        obs=mk(T, N, K0), next_obs=mk(T, N, K0), reward=mk(T, N),
        done=torch.empty(T, N, dtype=torch.bool, device=dev),
        truncated=torch.empty(T, N, dtype=torch.bool, device=dev),
        raw=mk(T, N, A), action=mk(T, N, A), log_likelihood=mk(T, N), mu=mk(T, N, A),
        sigma=mk(T, N, A), value=mk(T, N, c_dims[-1]), h_out=mk(N, H),
        key_out=torch.empty_like(env_key), count_out=torch.empty_like(env_step_count),
        counter_out=torch.empty_like(wrap_step_counter), obs_out=mk(N, K0), reward_out=mk(N))
    P = ctypes.c_void_p * Lc
    i64s = lambda v: (ctypes.c_int64 * len(v))(*[int(q) for q in v])
    n_mean, n_m2, n_cnt, n_eps = norm if norm is not None else (None, None, None, 0.0)
    if profiler.active:
        flop = K0 * H + 6 * H * H + H * A2 + sum(c_dims[l] * c_dims[l + 1] for l in range(Lc))
        profiler.next_flops = 2.0 * T * N * flop
        profiler.next_bytes = 4.0 * T * N * (2 * K0 + 4 * A + 2 + c_dims[-1]) + 2.0 * T * N
    check(lib().mi_rollout_mock_gru_ws_bf16(
        ptr(env_key, i64), ptr(env_step_count, i64), ptr(wrap_step_counter, i64), ptr(obs0, f32),
        ptr(reset_key.reshape(1), i64), int(max_steps), int(max_len), int(T), N, K0, H, A2,
        ptr(n_mean, f32), ptr(n_m2, f32), ptr(n_cnt, f32), float(n_eps),
        ptr(dense_in[0], bf16), ptr(dense_in[1], f32), ptr(proj[0], bf16), ptr(proj[1], f32),
        ptr(w_h, f32), ptr(b_hn, f32), ptr(dense_out[0], bf16), ptr(dense_out[1], f32),
        ptr(h_in, f32), ptr(out["h_out"], f32),
        Lc, P(*[ptr(t) for t in c_w]), P(*[ptr(t) for t in c_b]), i64s(c_dims), i64s(c_acts),
        ptr(rng_state), int(offset_add), float(min_std), float(std_scale), float(entropy_weight),
        int(bool(deterministic)), ptr(out["obs"], f32), ptr(out["next_obs"], f32),
        ptr(out["reward"], f32), ptr(out["done"].view(u8)), ptr(out["truncated"].view(u8)),
        ptr(out["raw"], f32), ptr(out["action"], f32), ptr(out["log_likelihood"], f32),
        ptr(out["mu"], f32), ptr(out["sigma"], f32), ptr(out["value"], f32),
        ptr(out["key_out"], i64), ptr(out["count_out"], i64), ptr(out["counter_out"], i64),
        ptr(out["obs_out"], f32), ptr(out["reward_out"], f32), stream()),
        "mi_rollout_mock_gru_ws_bf16")
    return out


def key_permutations(key: torch.Tensor, n_perm: int, n: int) -> torch.Tensor:
    """int64 [n_perm, n]: row e = random.permutation(random.fold_in(key, e), n)."""
    _need(key.dtype == i64 and key.numel() == 1, "key_permutations: one int64 key")
    _need(0 <= n <= 8192, "key_permutations: n <= 8192")
    out = torch.empty(n_perm, n, dtype=i64, device=key.device)
    check(lib().mi_key_permutations(ptr(key.reshape(1), i64), ptr(out, i64), int(n_perm), int(n),
                                    stream()), "mi_key_permutations")
    return out


def key_fold(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _need(a.dtype == i64 and b.dtype == i64 and a.shape == b.shape, "key_fold: int64 same shape")
    a = a if a.is_contiguous() else a.contiguous()
    b = b if b.is_contiguous() else b.contiguous()
    out = torch.empty_like(a)
    check(lib().mi_key_fold(ptr(a, i64), ptr(b, i64), ptr(out, i64), a.numel(), stream()),
          "mi_key_fold")
    return out


def episode_step(counter: torch.Tensor, inner_done: torch.Tensor, inner_trunc, max_len: int):
    """Returns (counter + 1, truncated (bool), done (float32), done (bool)) —
    episode_wrapper.py:12-22."""
    n = counter.numel()
    _need(counter.dtype == i64 and inner_done.numel() == n, "episode_step: shapes")
    is_float = inner_done.dtype == f32
    d = inner_done if is_float else _as_u8(inner_done)
    t = None if inner_trunc is None else _as_u8(inner_trunc)
    c_out = torch.empty_like(counter)
    t_out = torch.empty(counter.shape, dtype=torch.bool, device=counter.device)
    d_out = torch.empty(counter.shape, dtype=f32, device=counter.device)
    f_out = torch.empty(counter.shape, dtype=torch.bool, device=counter.device)
    check(lib().mi_episode_step(ptr(counter, i64), ptr(d), int(is_float), ptr(t), int(max_len),
                                ptr(c_out, i64), ptr(t_out.view(torch.uint8)), ptr(d_out, f32),
                                ptr(f_out.view(torch.uint8)), n, stream()), "mi_episode_step")
    return c_out, t_out, d_out, f_out


def episode_step_select(counter: torch.Tensor, inner_done: torch.Tensor, inner_trunc,
                        max_len: int, reset_counter: torch.Tensor, reset_trunc: torch.Tensor,
                        reset_done: torch.Tensor, pairs: list, producer=None):
    """`episode_step` and the reset-on-done select of every (on_true, on_false) leaf of
    `pairs` (<= 16) in ONE launch (`mi_episode_step_select`).  Returns
    (counter', truncated, done, done flag, selected counter / truncated / done, outs).
    `producer` (envs/synthetic.py: MockEnv.step_deferred): the inner env's step runs inside
    the same launch (`mi_mock_episode_step_select`); `inner_done` is then unused."""
    B = counter.numel()
    _need(counter.dtype == i64 and counter.dim() == 1
          and (producer is not None or inner_done.numel() == B), "episode_step_select: shapes")
    _need(reset_counter.dtype == i64 and reset_trunc.dtype == torch.bool
          and reset_done.dtype == f32 and reset_counter.shape == counter.shape
          and reset_trunc.shape == counter.shape and reset_done.shape == counter.shape,
          "episode_step_select: reset leaves must be int64 / bool / float32 [B]")
    _need(len(pairs) <= 16, "episode_step_select: at most 16 leaves")
    is_float = producer is None and inner_done.dtype == f32
    d_in = None if producer is not None else (inner_done if is_float else _as_u8(inner_done))
    t_in = None if inner_trunc is None else _as_u8(inner_trunc)
    dev = counter.device
    mk = lambda dt: torch.empty(counter.shape, dtype=dt, device=dev)
    c_out, c_sel = mk(i64), mk(i64)
    t_out, t_sel, f_out = mk(torch.bool), mk(torch.bool), mk(torch.bool)
    d_out, d_sel = mk(f32), mk(f32)
    v = lambda t: t.view(torch.uint8) if t.dtype == torch.bool else t
    outs, tab = [], []
    for on_true, on_false in pairs:
        _need(on_false.shape[0] == B and on_true.dtype == on_false.dtype
              and on_true.is_contiguous() and on_false.is_contiguous(),
              "episode_step_select: leaf layout")
        row_bytes = on_false.element_size()
        for dd in on_false.shape[1:]:
            row_bytes *= dd
        if on_true.shape == on_false.shape:
            stride = row_bytes
        elif on_true.shape == on_false.shape[1:]:
            stride = 0
        else:
            raise MippoError("episode_step_select: on_true must match on_false or be one row")
        out = torch.empty_like(on_false)
        outs.append(out)
        kind, col0 = (0, 0) if producer is None else producer["leaves"].get(id(on_false), (0, 0))
        if row_bytes:
            tab.append((ptr(v(on_true)), stride, ptr(v(on_false)), ptr(v(out)), row_bytes,
                        kind, col0))
        else:
            _need(kind == 0, "episode_step_select: empty produced leaf")
    n = len(tab)
    P = ctypes.c_void_p * max(n, 1)
    L = ctypes.c_int64 * max(n, 1)
    col = lambda k: [r[k] for r in tab] or [None if k in (0, 2, 3) else 0]
    if producer is not None:
        _need(sum(1 for r in tab if r[5]) == len(producer["leaves"]),
              "episode_step_select: a produced leaf is not among the selected leaves")
        check(lib().mi_mock_episode_step_select(
            ptr(producer["key"], i64), ptr(producer["count"], i64), int(producer["max_steps"]),
            L(*col(5)), L(*col(6)), ptr(counter, i64), ptr(t_in), int(max_len), ptr(c_out, i64),
            ptr(t_out.view(torch.uint8)), ptr(d_out, f32), ptr(f_out.view(torch.uint8)),
            ptr(reset_counter, i64), ptr(reset_trunc.view(torch.uint8)), ptr(reset_done, f32),
            ptr(c_sel, i64), ptr(t_sel.view(torch.uint8)), ptr(d_sel, f32),
            P(*col(0)), L(*col(1)), P(*col(2)), P(*col(3)), L(*col(4)), n, B, stream()),
            "mi_mock_episode_step_select")
        return c_out, t_out, d_out, f_out, c_sel, t_sel, d_sel, outs
    check(lib().mi_episode_step_select(
        ptr(counter, i64), ptr(d_in), int(is_float), ptr(t_in), int(max_len), ptr(c_out, i64),
        ptr(t_out.view(torch.uint8)), ptr(d_out, f32), ptr(f_out.view(torch.uint8)),
        ptr(reset_counter, i64), ptr(reset_trunc.view(torch.uint8)), ptr(reset_done, f32),
        ptr(c_sel, i64), ptr(t_sel.view(torch.uint8)), ptr(d_sel, f32),
        P(*col(0)), L(*col(1)), P(*col(2)), P(*col(3)), L(*col(4)), n, B, stream()),
        "mi_episode_step_select")
    return c_out, t_out, d_out, f_out, c_sel, t_sel, d_sel, outs


# ------------------------------------------------------------- a20: GRU
def gru_mfma_ok(H: int) -> bool:
    """Hidden sizes the matrix-core GRU kernels take (csrc/gru_mfma.hip)."""
    return H in (32, 64, 96, 128)


def gru_seq_fwd(gi, w_h, b_hn, h0, done, train: bool, mfma: bool = False):
    """gi [T,B,3H] -> (h_out [T,B,H], h_prev | None, gates | None, h_final [B,H]).
    `mfma`: h W_h on the bf16 matrix cores (`mi_gru_seq_fwd_bf16`); training then also
    leaves the bf16 image of h_prev (the dW operand) in `h_prev.bf16_image` [T*B, H]."""
    T, B, H3 = gi.shape
    H = H3 // 3
    _need(w_h.shape == (H, H3) and b_hn.shape == (H,) and h0.shape == (B, H), "gru_seq_fwd: shapes")
    dev = gi.device
    h_out = torch.empty(T, B, H, dtype=f32, device=dev)
    h_prev = torch.empty(T, B, H, dtype=f32, device=dev) if train else None
    gates = torch.empty(T, B, 4 * H, dtype=f32, device=dev) if train else None
    h_final = torch.empty(B, H, dtype=f32, device=dev)
    d = None if done is None else _as_u8(done)
    if d is not None:
        _need(d.shape == (T, B), "gru_seq_fwd: done must be [T, B]")
    if mfma:
        hp_bf = torch.empty(T * B, H, dtype=bf16, device=dev) if (train and H % 8 == 0) else None
        check(lib().mi_gru_seq_fwd_bf16(
            ptr(gi, f32), ptr(w_h, f32), ptr(b_hn, f32), ptr(h0, f32), ptr(d), ptr(h_out, f32),
            ptr(h_prev, f32), ptr(gates, f32), ptr(h_final, f32), ptr(hp_bf), T, B, H, stream()),
            "mi_gru_seq_fwd_bf16")
        if hp_bf is not None:
            h_prev.bf16_image = hp_bf
        return h_out, h_prev, gates, h_final
    check(lib().mi_gru_seq_fwd_f32(
        ptr(gi, f32), ptr(w_h, f32), ptr(b_hn, f32), ptr(h0, f32), ptr(d), ptr(h_out, f32),
        ptr(h_prev, f32), ptr(gates, f32), ptr(h_final, f32), T, B, H, stream()),
        "mi_gru_seq_fwd_f32")
    return h_out, h_prev, gates, h_final


def gru_seq_fwd_tail_supported(T: int, H: int, N_out: int) -> bool:
    return bool(lib().mi_gru_seq_fwd_tail_supported(int(T), int(H), int(N_out)))


def gru_seq_fwd_tail(gi, w_h, b_hn, h0, done, w_out_ff, b_out, N_out: int, extras, rng_state,
                     offset_add: int, *, min_std: float, std_scale: float,
                     entropy_weight: float, eps2=None):
    """`gru_seq_fwd(train=True, mfma=True)` with the head Dense(H -> N_out) and the sampler's
    replay (stored raw actions `extras` [T*B, A] scored) inside the launch
    (`mi_gru_seq_fwd_tail_bf16`).  Returns (h_out, h_prev (with `.bf16_image`), gates, h_final,
    ms [T*B, N_out], h_bf [T*B, H], log_likelihood [T*B], reg [T*B])."""
    T, B, H3 = gi.shape
    H = H3 // 3
    _need(w_h.shape == (H, H3) and b_hn.shape == (H,) and h0.shape == (B, H),
          "gru_seq_fwd_tail: shapes")
    dev = gi.device
    M = T * B
    h_out = torch.empty(T, B, H, dtype=f32, device=dev)
    h_prev = torch.empty(T, B, H, dtype=f32, device=dev)
    gates = torch.empty(T, B, 4 * H, dtype=f32, device=dev)
    h_final = torch.empty(B, H, dtype=f32, device=dev)
    hp_bf = torch.empty(M, H, dtype=bf16, device=dev)
    h_bf = torch.empty(M, H, dtype=bf16, device=dev)
    ms = torch.empty(M, N_out, dtype=f32, device=dev)
    ll = torch.empty(M, dtype=f32, device=dev)
    reg = torch.empty(M, dtype=f32, device=dev)
    d = None if done is None else _as_u8(done)
    _need(extras.shape == (M, N_out // 2) and extras.is_contiguous(),
          "gru_seq_fwd_tail: extras must be a contiguous [T*B, A]")
    check(lib().mi_gru_seq_fwd_tail_bf16(
        ptr(gi, f32), ptr(w_h, f32), ptr(b_hn, f32), ptr(h0, f32), ptr(d), ptr(h_out, f32),
        ptr(h_prev, f32), ptr(gates, f32), ptr(h_final, f32), ptr(hp_bf), ptr(w_out_ff, bf16),
        ptr(b_out, f32), int(N_out), ptr(ms, f32), ptr(h_bf), ptr(extras, f32), ptr(rng_state),
        int(offset_add), ptr(eps2, f32), float(min_std), float(std_scale),
        float(entropy_weight), ptr(ll, f32), ptr(reg, f32), T, B, H, stream()),
        "mi_gru_seq_fwd_tail_bf16")
    h_prev.bf16_image = hp_bf
    return h_out, h_prev, gates, h_final, ms, h_bf, ll, reg


def gru_seq_proj_supported(T: int, B: int, H: int, K_in: int, N_out: int) -> bool:
    return bool(lib().mi_gru_seq_proj_supported(int(T), int(B), int(H), int(K_in), int(N_out)))


def gru_seq_fwd_proj_tail(y_bf, w_i_ff, b_i, w_h, b_hn, h0, done, w_out_ff, b_out, N_out: int,
                          extras, rng_state, offset_add: int, T: int, *, min_std: float,
                          std_scale: float, entropy_weight: float, eps2=None):
    """`gru_seq_fwd_tail` with gi = y W_i + b_i evaluated inside the launch from `y_bf`
    [T*B, H], the bf16 image of the GRU's input (`mi_gru_seq_fwd_proj_tail_bf16`).  Same
    returns."""
    M, H = y_bf.shape
    B = M // T
    _need(w_h.shape == (H, 3 * H) and b_hn.shape == (H,) and h0.shape == (B, H)
          and b_i.shape == (3 * H,) and y_bf.dtype == bf16 and y_bf.is_contiguous(),
          "gru_seq_fwd_proj_tail: shapes")
    dev = y_bf.device
    h_out = torch.empty(T, B, H, dtype=f32, device=dev)
    h_prev = torch.empty(T, B, H, dtype=f32, device=dev)
    gates = torch.empty(T, B, 4 * H, dtype=f32, device=dev)
    h_final = torch.empty(B, H, dtype=f32, device=dev)
    hp_bf = torch.empty(M, H, dtype=bf16, device=dev)
    h_bf = torch.empty(M, H, dtype=bf16, device=dev)
    ms = torch.empty(M, N_out, dtype=f32, device=dev)
    ll = torch.empty(M, dtype=f32, device=dev)
    reg = torch.empty(M, dtype=f32, device=dev)
    d = None if done is None else _as_u8(done)
    _need(extras.shape == (M, N_out // 2) and extras.is_contiguous(),
          "gru_seq_fwd_proj_tail: extras must be a contiguous [T*B, A]")
    check(lib().mi_gru_seq_fwd_proj_tail_bf16(
        ptr(y_bf), H, ptr(w_i_ff, bf16), ptr(b_i, f32), ptr(w_h, f32), ptr(b_hn, f32),
        ptr(h0, f32), ptr(d), ptr(h_out, f32), ptr(h_prev, f32), ptr(gates, f32),
        ptr(h_final, f32), ptr(hp_bf), ptr(w_out_ff, bf16), ptr(b_out, f32), int(N_out),
        ptr(ms, f32), ptr(h_bf), ptr(extras, f32), ptr(rng_state), int(offset_add),
        ptr(eps2, f32), float(min_std), float(std_scale), float(entropy_weight), ptr(ll, f32),
        ptr(reg, f32), T, B, H, stream()), "mi_gru_seq_fwd_proj_tail_bf16")
    h_prev.bf16_image = hp_bf
    return h_out, h_prev, gates, h_final, ms, h_bf, ll, reg


def gru_seq_front_supported(T: int, B: int, H: int, K0: int, N_out: int) -> bool:
    return bool(lib().mi_gru_seq_front_supported(int(T), int(B), int(H), int(K0), int(N_out)))


def gru_seq_fwd_front_proj_tail(x2, w0_ff, b0, w_i_ff, b_i, w_h, b_hn, h0, done, w_out_ff, b_out,
                                N_out: int, extras, rng_state, offset_add: int, T: int, *,
                                min_std: float, std_scale: float, entropy_weight: float,
                                eps2=None):
    """`gru_seq_fwd_proj_tail` with the relu Dense(K0 <= 8 -> H) in front inside the launch too
    (`mi_gru_seq_fwd_front_proj_tail_bf16`): x2 [T*B, K0] fp32 is that layer's input.  Returns
    gru_seq_fwd_tail's tuple + (x_bf [T*B, 8], y_bf [T*B, H]), the layer's bf16 images."""
    M, K0 = x2.shape
    H = w_h.shape[0]
    B = M // T
    _need(w_h.shape == (H, 3 * H) and b_hn.shape == (H,) and h0.shape == (B, H)
          and b_i.shape == (3 * H,) and b0.shape == (H,) and 1 <= K0 <= 8 and x2.is_contiguous()
          and x2.dtype == f32, "gru_seq_fwd_front_proj_tail: shapes")
    dev = x2.device
    h_out = torch.empty(T, B, H, dtype=f32, device=dev)
    h_prev = torch.empty(T, B, H, dtype=f32, device=dev)
    gates = torch.empty(T, B, 4 * H, dtype=f32, device=dev)
    h_final = torch.empty(B, H, dtype=f32, device=dev)
    hp_bf = torch.empty(M, H, dtype=bf16, device=dev)
    h_bf = torch.empty(M, H, dtype=bf16, device=dev)
    x_bf = torch.empty(M, 8, dtype=bf16, device=dev)
    y_bf = torch.empty(M, H, dtype=bf16, device=dev)
    ms = torch.empty(M, N_out, dtype=f32, device=dev)
    ll = torch.empty(M, dtype=f32, device=dev)
    reg = torch.empty(M, dtype=f32, device=dev)
    d = None if done is None else _as_u8(done)
    _need(extras.shape == (M, N_out // 2) and extras.is_contiguous(),
          "gru_seq_fwd_front_proj_tail: extras must be a contiguous [T*B, A]")
    check(lib().mi_gru_seq_fwd_front_proj_tail_bf16(
        ptr(x2, f32), K0, ptr(w0_ff, bf16), ptr(b0, f32), ptr(x_bf), ptr(y_bf), ptr(w_i_ff, bf16),
        ptr(b_i, f32), ptr(w_h, f32), ptr(b_hn, f32), ptr(h0, f32), ptr(d), ptr(h_out, f32),
        ptr(h_prev, f32), ptr(gates, f32), ptr(h_final, f32), ptr(hp_bf), ptr(w_out_ff, bf16),
        ptr(b_out, f32), int(N_out), ptr(ms, f32), ptr(h_bf), ptr(extras, f32), ptr(rng_state),
        int(offset_add), ptr(eps2, f32), float(min_std), float(std_scale), float(entropy_weight),
        ptr(ll, f32), ptr(reg, f32), T, B, H, stream()), "mi_gru_seq_fwd_front_proj_tail_bf16")
    h_prev.bf16_image = hp_bf
    return h_out, h_prev, gates, h_final, ms, h_bf, ll, reg, x_bf, y_bf


def gru_seq_bwd_proj_tail(y_bf, w_i_fb, gates, h_prev, w_h, done, w_out_fb, N_out: int,
                          mean_and_std, extras, rng_state, offset_add: int, g_ll, g_reg: float, *,
                          min_std: float, std_scale: float, entropy_weight: float, eps2=None):
    """`gru_seq_bwd_tail` with the input projection's backward inside
    (`mi_gru_seq_bwd_proj_tail_bf16`).  Returns the bf16 images (dgi [T*B, 3H], dz0 [T*B, H],
    dgh [T*B, 3H], dz_out [T*B, pad8(N_out)])."""
    T, B, H = h_prev.shape
    dev = h_prev.device
    M = T * B
    _need(y_bf.shape == (M, H) and y_bf.dtype == bf16 and y_bf.is_contiguous(),
          "gru_seq_bwd_proj_tail: y_bf must be the contiguous [T*B, H] image")
    _need(mean_and_std.shape == (M, N_out) and extras.shape == (M, N_out // 2),
          "gru_seq_bwd_proj_tail: sampler operands must be [T*B, N_out] / [T*B, A]")
    if g_ll is not None:
        _need(g_ll.shape == (M,) and g_ll.is_contiguous(),
              "gru_seq_bwd_proj_tail: g_ll must be [T*B]")
    dgi_bf = torch.empty(M, 3 * H, dtype=bf16, device=dev)
    dz0_bf = torch.empty(M, H, dtype=bf16, device=dev)
    dgh_bf = torch.empty(M, 3 * H, dtype=bf16, device=dev)
    dz_bf = torch.empty(M, (N_out + 7) // 8 * 8, dtype=bf16, device=dev)
    d = None if done is None else _as_u8(done)
    check(lib().mi_gru_seq_bwd_proj_tail_bf16(
        ptr(y_bf), H, ptr(w_i_fb, bf16), ptr(dgi_bf), ptr(dz0_bf), ptr(gates, f32),
        ptr(h_prev, f32), ptr(w_h, f32), ptr(d), None, ptr(dgh_bf), ptr(w_out_fb, bf16),
        int(N_out), ptr(mean_and_std, f32), ptr(extras, f32), ptr(rng_state), int(offset_add),
        ptr(eps2, f32), ptr(g_ll, f32), float(g_reg), float(min_std), float(std_scale),
        float(entropy_weight), ptr(dz_bf), T, B, H, stream()), "mi_gru_seq_bwd_proj_tail_bf16")
    return dgi_bf, dz0_bf, dgh_bf, dz_bf


def gru_seq_bwd_tail_supported(T: int, H: int, N_out: int) -> bool:
    return bool(lib().mi_gru_seq_bwd_tail_supported(int(T), int(H), int(N_out)))


def gru_seq_bwd_tail(gates, h_prev, w_h, done, w_out_fb, N_out: int, mean_and_std, extras,
                     rng_state, offset_add: int, g_ll, g_reg: float, *, min_std: float,
                     std_scale: float, entropy_weight: float, eps2=None):
    """`gru_seq_bwd(mfma=True, dgh_as_bf16=True)` with the sampler's backward and the head's dX
    in front of the BPTT inside the launch (`mi_gru_seq_bwd_tail_bf16`).  Returns
    (dgi [T,B,3H], dgh bf16 image [T*B, 3H], dz_out bf16 image [T*B, pad8(N_out)] — the
    head's dW operand)."""
    T, B, H = h_prev.shape
    dev = h_prev.device
    M = T * B
    _need(mean_and_std.shape == (M, N_out) and extras.shape == (M, N_out // 2),
          "gru_seq_bwd_tail: sampler operands must be [T*B, N_out] / [T*B, A]")
    if g_ll is not None:
        _need(g_ll.shape == (M,) and g_ll.is_contiguous(), "gru_seq_bwd_tail: g_ll must be [T*B]")
    dgi = torch.empty(T, B, 3 * H, dtype=f32, device=dev)
    dgh_bf = torch.empty(M, 3 * H, dtype=bf16, device=dev)
    dz_bf = torch.empty(M, (N_out + 7) // 8 * 8, dtype=bf16, device=dev)
    d = None if done is None else _as_u8(done)
    check(lib().mi_gru_seq_bwd_tail_bf16(
        ptr(gates, f32), ptr(h_prev, f32), ptr(w_h, f32), ptr(d), ptr(dgi, f32), None,
        ptr(dgh_bf), ptr(w_out_fb, bf16), int(N_out), ptr(mean_and_std, f32), ptr(extras, f32),
        ptr(rng_state), int(offset_add), ptr(eps2, f32), ptr(g_ll, f32), float(g_reg),
        float(min_std), float(std_scale), float(entropy_weight), ptr(dz_bf), T, B, H, stream()),
        "mi_gru_seq_bwd_tail_bf16")
    return dgi, dgh_bf, dz_bf


def gru_seq_bwd(g_h, gates, h_prev, w_h, done, mfma: bool = False, dgh_as_bf16: bool = False,
                dh0_out=None):
    """Returns (dgi [T,B,3H], dgh [T,B,3H]).  `dgh_as_bf16` (matrix-core path): dgh comes
    back as its bf16 image [T*B, 3H] — the dz operand of the recurrent kernel's dW launch —
    and the fp32 tensor is not written.  `dh0_out` [B, H] (optional) receives the gradient
    w.r.t. the sequence's first carry."""
    T, B, H = g_h.shape
    dev = g_h.device
    dgi = torch.empty(T, B, 3 * H, dtype=f32, device=dev)
    d = None if done is None else _as_u8(done)
    if dh0_out is not None:
        _need(dh0_out.shape == (B, H) and dh0_out.is_contiguous(), "gru_seq_bwd: dh0_out [B, H]")
    if mfma:
        as_bf = dgh_as_bf16 and (3 * H) % 8 == 0
        dgh = None if as_bf else torch.empty(T, B, 3 * H, dtype=f32, device=dev)
        dgh_bf = torch.empty(T * B, 3 * H, dtype=bf16, device=dev) if as_bf else None
        check(lib().mi_gru_seq_bwd_bf16(
            ptr(g_h, f32), ptr(gates, f32), ptr(h_prev, f32), ptr(w_h, f32), ptr(d),
            ptr(dgi, f32), ptr(dgh, f32), ptr(dh0_out, f32), ptr(dgh_bf), T, B, H, stream()),
            "mi_gru_seq_bwd_bf16")
        return dgi, (dgh_bf if as_bf else dgh)
    dgh = torch.empty(T, B, 3 * H, dtype=f32, device=dev)
    check(lib().mi_gru_seq_bwd_f32(
        ptr(g_h, f32), ptr(gates, f32), ptr(h_prev, f32), ptr(w_h, f32), ptr(d),
        ptr(dgi, f32), ptr(dgh, f32), ptr(dh0_out, f32), T, B, H, stream()), "mi_gru_seq_bwd_f32")
    return dgi, dgh


# ------------------------------------------------------------- f2: LSTM
ACT_SIGMOID = 4  # LSTM gate functions only (MI_ACT_SIGMOID)


def lstm_seq_fwd(gi, w_h, h0, c0, done, train: bool, mfma: bool = False, h_init=None,
                 c_init=None, gate_act: int = ACT_SIGMOID, cell_act: int = ACT_TANH):
    """gi [T,B,4H] -> (h_out [T,B,H], h_prev | None, c_prev | None, gates [T,B,5H] | None,
    h_final [B,H], c_final [B,H]).  `mfma`: h W_h on the bf16 matrix cores (default gate
    functions and zero reset only).  `h_init` / `c_init` [H]: the carry a done row is reset
    to (the learnable initial state); `gate_act` / `cell_act`: MI_ACT codes."""
    T, B, H4 = gi.shape
    H = H4 // 4
    _need(w_h.shape == (H, H4) and h0.shape == (B, H) and c0.shape == (B, H),
          "lstm_seq_fwd: shapes")
    dev = gi.device
    mk = lambda *s: torch.empty(*s, dtype=f32, device=dev)
    h_out = mk(T, B, H)
    h_prev = mk(T, B, H) if train else None
    c_prev = mk(T, B, H) if train else None
    gates = mk(T, B, 5 * H) if train else None
    h_final, c_final = mk(B, H), mk(B, H)
    d = None if done is None else _as_u8(done)
    if d is not None:
        _need(d.shape == (T, B), "lstm_seq_fwd: done must be [T, B]")
    if mfma:
        _need(h_init is None and c_init is None and gate_act == ACT_SIGMOID
              and cell_act == ACT_TANH, "lstm_seq_fwd: the matrix-core recurrence takes the "
              "default gate functions and a zero reset")
        check(lib().mi_lstm_seq_fwd_bf16(
            ptr(gi, f32), ptr(w_h, f32), ptr(h0, f32), ptr(c0, f32), ptr(d), ptr(h_out, f32),
            ptr(h_prev, f32), ptr(c_prev, f32), ptr(gates, f32), ptr(h_final, f32),
            ptr(c_final, f32), T, B, H, stream()), "mi_lstm_seq_fwd_bf16")
    else:
        _need((h_init is None) == (c_init is None), "lstm_seq_fwd: h_init and c_init go together")
        _need(h_init is None or (h_init.shape == (H,) and c_init.shape == (H,)),
              "lstm_seq_fwd: h_init / c_init must be [H]")
        check(lib().mi_lstm_seq_fwd_f32(
            ptr(gi, f32), ptr(w_h, f32), ptr(h0, f32), ptr(c0, f32), ptr(d), ptr(h_out, f32),
            ptr(h_prev, f32), ptr(c_prev, f32), ptr(gates, f32), ptr(h_final, f32),
            ptr(c_final, f32), ptr(h_init, f32), ptr(c_init, f32), int(gate_act), int(cell_act),
            T, B, H, stream()), "mi_lstm_seq_fwd_f32")
    return h_out, h_prev, c_prev, gates, h_final, c_final


def lstm_seq_bwd(g_h, gates, c_prev, w_h, done, mfma: bool = False, want_dinit: bool = False,
                 gate_act: int = ACT_SIGMOID, cell_act: int = ACT_TANH):
    """Returns d_gates [T,B,4H] (gradient w.r.t. the gate pre-activations), and — with
    `want_dinit` — the gradients (d h_init [H], d c_init [H]) of a learnable initial state."""
    T, B, H = g_h.shape
    da = torch.empty(T, B, 4 * H, dtype=f32, device=g_h.device)
    d = None if done is None else _as_u8(done)
    if mfma:
        _need(not want_dinit and gate_act == ACT_SIGMOID and cell_act == ACT_TANH,
              "lstm_seq_bwd: the matrix-core BPTT takes the default gate functions only")
        check(lib().mi_lstm_seq_bwd_bf16(
            ptr(g_h, f32), ptr(gates, f32), ptr(c_prev, f32), ptr(w_h, f32), ptr(d),
            ptr(da, f32), None, None, T, B, H, stream()), "mi_lstm_seq_bwd_bf16")
        return da
    part = None
    if want_dinit:
        G = lib().mi_lstm_seq_bwd_blocks(B, H)
        _need(G >= 1, "lstm_seq_bwd: bad shape")
        part = torch.empty(G, 2, H, dtype=f32, device=g_h.device)
    check(lib().mi_lstm_seq_bwd_f32(
        ptr(g_h, f32), ptr(gates, f32), ptr(c_prev, f32), ptr(w_h, f32), ptr(d), ptr(da, f32),
        None, None, ptr(part, f32), int(gate_act), int(cell_act), T, B, H, stream()),
        "mi_lstm_seq_bwd_f32")
    if want_dinit:
        dinit = part.sum(dim=0)  # block order: deterministic
        return da, dinit[0], dinit[1]
    return da
