"""Integer key derivation for env resets, minibatch permutations and seeds.

The reference threads `jax.random` keys (`ppo.py:271,284-294,544-548`,
`rollout.py:57-59`, `episode_wrapper.py:26-30`).  JAX's threefry streams cannot
be reproduced without JAX (parity on RNG *values* is unpinned, SURVEY §7 hard
part 3), so keys here are int64 tensors mixed with splitmix64.  Everything is
plain integer torch arithmetic (wrap-around multiply, xor, logical shift), so
the same key gives bit-identical children, integers and permutations on CPU
and on the GPU — index parity between the oracle run and the HIP run is exact.
"""
from __future__ import annotations

import torch


def _ops():
    from . import ops  # deferred: ops imports torch + the C ABI binding

    return ops


# The key functions below run as one kernel launch each on the GPU (csrc/keys.hip) and as
# plain integer torch arithmetic on the CPU — the same bits either way.  Inside
# `torch.func.vmap` (envs/vmap_env.py lifts single-env code with it) a kernel launch is
# not possible (a batched tensor has no device pointer), so `torch_only()` routes GPU
# tensors through the torch arithmetic too.
_TORCH_ONLY = [0]


class torch_only:
    def __enter__(self):
        _TORCH_ONLY[0] += 1
        return self

    def __exit__(self, *exc):
        _TORCH_ONLY[0] -= 1
        return False


def _kernel(k: torch.Tensor) -> bool:
    return k.is_cuda and not _TORCH_ONLY[0]


_GOLDEN = -7046029254386353131  # 0x9E3779B97F4A7C15 as int64
_M1 = -4658895280553007687  # 0xBF58476D1CE4E5B9
_M2 = -7723592293110705685  # 0x94D049BB133111EB


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    """Logical shift right on int64 (torch's >> is arithmetic)."""
    return (x >> s) & ((1 << (64 - s)) - 1)


def _mix(z: torch.Tensor) -> torch.Tensor:
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def key(seed: int, device=None) -> torch.Tensor:
    """Scalar int64 key from an integer seed (`jax.random.key`)."""
    s = int(seed) & 0xFFFFFFFFFFFFFFFF
    if s >= 1 << 63:
        s -= 1 << 64
    return _mix(torch.tensor(s, dtype=torch.int64, device=device) + _GOLDEN)


def split(k: torch.Tensor, num=2) -> torch.Tensor:
    """Children of `k` (any shape): result shape `k.shape + shape(num)`
    (`jax.random.split(key, num)`; `num` may be an int or a shape tuple)."""
    shape = (num,) if isinstance(num, int) else tuple(num)
    n = 1
    for s in shape:
        n *= s
    if _kernel(k):  # same integers, one launch (csrc/keys.hip)
        return _ops().key_expand(k, n, 0).reshape((*k.shape, *shape))
    idx = torch.arange(1, n + 1, dtype=torch.int64, device=k.device)
    out = _mix(k.unsqueeze(-1) + idx * _GOLDEN)
    return out.reshape((*k.shape, *shape))


def split2(k: torch.Tensor):
    """`a, b = split(k)` as two CONTIGUOUS tensors of k's shape (one launch on the
    GPU; same integers as `split(k)[..., 0]`, `split(k)[..., 1]`)."""
    if _kernel(k):
        out = _ops().key_expand(k, 2, 0, child_major=True)
        return out[0], out[1]
    s = split(k, 2)
    return s[..., 0].contiguous(), s[..., 1].contiguous()


def _to_i64(v: int) -> int:
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def _mix_host(z: int) -> int:
    """`_mix` on a Python int (same bits as the tensor version)."""
    m = 0xFFFFFFFFFFFFFFFF
    z &= m
    z = ((z ^ (z >> 30)) * (_M1 & m)) & m
    z = ((z ^ (z >> 27)) * (_M2 & m)) & m
    return z ^ (z >> 31)


def fold_in(k: torch.Tensor, data: int) -> torch.Tensor:
    """`jax.random.fold_in`: a new key from a key and an integer.  The integer is
    mixed on the host and enters as a scalar operand, so no host-to-device copy
    is issued (the call is legal inside HIP-graph capture)."""
    c = _to_i64(_mix_host(int(data) + _GOLDEN))
    return _mix(k ^ c)


def bits(k: torch.Tensor, shape=()) -> torch.Tensor:
    """64 random bits per element: result `k.shape + shape` (int64)."""
    shape = tuple(shape)
    n = 1
    for s in shape:
        n *= s
    if _kernel(k):
        return _ops().key_expand(k, n, 1).reshape((*k.shape, *shape))
    idx = torch.arange(1, n + 1, dtype=torch.int64, device=k.device)
    out = _mix(_mix(k).unsqueeze(-1) ^ (idx * _M2))
    return out.reshape((*k.shape, *shape))


def randint(k: torch.Tensor, shape, minval: int, maxval: int) -> torch.Tensor:
    """Integers in [minval, maxval) (`jax.random.randint`); int64."""
    span = int(maxval) - int(minval)
    if span <= 0:
        return torch.full((*k.shape, *tuple(shape)), int(minval), dtype=torch.int64,
                          device=k.device)
    if _kernel(k):
        return _ops().key_expand(k, _numel(shape), 2, minval, maxval).reshape(
            (*k.shape, *tuple(shape)))
    b = _lsr(bits(k, shape), 1)  # non-negative 63-bit
    return b % span + int(minval)


def uniform(k: torch.Tensor, shape=(), dtype=torch.float32) -> torch.Tensor:
    """U[0,1) with 24 random bits — exact in fp32, identical on CPU and GPU."""
    if _kernel(k) and dtype == torch.float32:
        return _ops().key_expand(k, _numel(shape), 3).reshape((*k.shape, *tuple(shape)))
    b = _lsr(bits(k, shape), 40)  # 24 bits
    return b.to(dtype) * (1.0 / (1 << 24))


def unit_uniform(k: torch.Tensor, shape=(), dtype=torch.float32,
                 fold: torch.Tensor | None = None) -> torch.Tensor:
    """Zero-mean unit-variance uniform noise, (u - 1/2)·sqrt(12).  One IEEE
    multiply after exact operands, so CPU and GPU agree bit for bit; synthetic
    envs use it where the reference's test envs draw `jax.random.normal`
    (`test_dummies/mock_env.py:41-52`).  `fold` (int64, k's shape): draw from
    `fold_key(k, fold)` — on the GPU in the same launch."""
    if _kernel(k) and dtype == torch.float32:
        return _ops().key_expand(k, _numel(shape), 4, fold=fold).reshape(
            (*k.shape, *tuple(shape)))
    if fold is not None:
        k = fold_key(k, fold)
    u = uniform(k, shape, dtype)
    return (u - 0.5) * 3.4641016151377544


def unit_normal(k: torch.Tensor, shape=(), fold: torch.Tensor | None = None) -> torch.Tensor:
    """N(0, 1) draws (`jax.random.normal`, the law of the reference's test envs,
    `test_dummies/mock_env.py:43,53`) by Box-Muller from the two 24-bit halves of each 64-bit
    hash: z = sqrt(-2 ln u1) cos(2 pi u2), u1 in (0, 1], u2 in [0, 1).  Plain torch ops on any
    device (no kernel of its own: the synthetic env's default law is `unit_uniform`, which
    the kernels evaluate bit for bit; this one agrees between CPU and GPU only to the
    transcendental functions' last ulp)."""
    if fold is not None:
        k = fold_key(k, fold)
    b = bits(k, shape)
    u1 = (_lsr(b, 40) + 1).to(torch.float32) * (1.0 / (1 << 24))
    u2 = (_lsr(b, 16) & 0xFFFFFF).to(torch.float32) * (1.0 / (1 << 24))
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(6.283185307179586 * u2)


def fold_key(k: torch.Tensor, data: torch.Tensor) -> torch.Tensor:
    """Per-element fold of an int64 tensor into keys of the same shape:
    mix(k ^ mix(data + GOLDEN))."""
    if _kernel(k):
        return _ops().key_fold(k, data)
    return _mix(k ^ _mix(data + _GOLDEN))


def _numel(shape) -> int:
    n = 1
    for s in tuple(shape):
        n *= s
    return n


def permutations(k: torch.Tensor, n_perm: int, n: int) -> torch.Tensor:
    """`stack([permutation(fold_in(k, e), n) for e in range(n_perm)])` — on the GPU (and
    n <= 8192) one launch for all of them."""
    if k.dim() != 0:
        raise ValueError("permutations expects a scalar key")
    if _kernel(k) and n <= 8192:
        return _ops().key_permutations(k, n_perm, n)
    return torch.stack([permutation(fold_in(k, e), n) for e in range(n_perm)], dim=0)


def permutation(k: torch.Tensor, n: int) -> torch.Tensor:
    """Random permutation of arange(n) (`jax.random.permutation(key, n)`):
    stable argsort of n distinct-with-overwhelming-probability 64-bit hashes."""
    if k.dim() != 0:
        raise ValueError("permutation expects a scalar key")
    h = bits(k, (n,))
    return torch.argsort(h, stable=True)
