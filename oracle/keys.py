"""Integer key derivation, restated independently of the product (numpy uint64).

TEST INFRASTRUCTURE ONLY (see `oracle/__init__.py`).

The reference threads `jax.random` keys through `ppo_step` (`nnx_ppo/algorithms/ppo.py:271,
284-294,544-548`, `rollout.py:57-59`, `wrappers/episode_wrapper.py:26-30`).  JAX's threefry
streams are a third-party dependency absent from the reference tree and from this image, so
the VALUES of the reference's random streams are PARITY UNPINNED; what the build pins is the
STRUCTURE (which key feeds what, `split` / `fold_in` / `permutation` call for call) on a key
scheme of its own: 64-bit keys mixed with SplitMix64 (Steele, Lea, Flood: "Fast splittable
pseudorandom number generators", OOPSLA 2014; the finaliser constants below are the published
ones, and `splitmix64_stream` reproduces the published reference outputs —
`tests/test_oracle_keys.py`).

This module states that scheme a second time, in numpy unsigned 64-bit arithmetic (natural
wrap-around, logical shifts), with none of the product's code: the product's CPU statement
(`nnx_ppo_amd/random.py`, torch int64 with emulated logical shifts) and its HIP kernels
(`csrc/keys.hip`) are both checked against it bit for bit, so index parity (reset keys,
minibatch permutations, observation noise) no longer rests on the product agreeing with itself.

API mirrors the functions the oracle's `ppo_step` takes through its `keys` argument; inputs
and outputs are torch int64 CPU tensors (two's-complement views of the uint64 values).
"""
from __future__ import annotations

import numpy as np
import torch

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)


def _u(t) -> np.ndarray:
    """uint64 view of an int64 tensor / array / Python int."""
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().contiguous().numpy().astype(np.int64, copy=False).view(np.uint64)
    if isinstance(t, np.ndarray):
        return t.astype(np.int64, copy=False).view(np.uint64)
    return np.array(int(t) & 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)


def _t(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.array(a, dtype=np.uint64, order="C").view(np.int64))


def mix(z: np.ndarray) -> np.ndarray:
    """The SplitMix64 finaliser (uint64 in, uint64 out)."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * M1
        z = (z ^ (z >> np.uint64(27))) * M2
        return z ^ (z >> np.uint64(31))


def splitmix64_stream(seed: int, n: int) -> list[int]:
    """The first n outputs of SplitMix64 seeded with `seed` (the published generator: state
    += GOLDEN, output = finaliser(state)) — for the known-answer test."""
    out, x = [], np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(n):
            x = x + GOLDEN
            out.append(int(mix(np.array(x, dtype=np.uint64))))
    return out


def key(seed: int) -> torch.Tensor:
    """`jax.random.key(seed)`: scalar key = finaliser(seed + GOLDEN)."""
    with np.errstate(over="ignore"):
        return _t(mix(np.array(np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF) + GOLDEN)))


def _count(num) -> tuple[tuple, int]:
    shape = (num,) if isinstance(num, int) else tuple(num)
    n = 1
    for s in shape:
        n *= s
    return shape, n


def split(k, num=2) -> torch.Tensor:
    """`jax.random.split(key, num)`: child i (1-based) of key k = finaliser(k + i GOLDEN);
    result shape k.shape + shape(num)."""
    shape, n = _count(num)
    ku = _u(k)
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        out = mix(ku[..., None] + idx * GOLDEN)
    return _t(out.reshape(ku.shape + shape))


def fold_in(k, data: int) -> torch.Tensor:
    """`jax.random.fold_in(key, int)`: finaliser(k xor finaliser(data + GOLDEN))."""
    with np.errstate(over="ignore"):
        c = mix(np.array(np.uint64(int(data) & 0xFFFFFFFFFFFFFFFF) + GOLDEN))
    return _t(mix(_u(k) ^ c))


def fold_key(k, data) -> torch.Tensor:
    """Element-wise fold of an integer tensor into keys of the same shape."""
    with np.errstate(over="ignore"):
        return _t(mix(_u(k) ^ mix(_u(data) + GOLDEN)))


def bits(k, shape=()) -> torch.Tensor:
    """64 bits per element: element i (1-based) = finaliser(finaliser(k) xor i M2)."""
    shape, n = _count(tuple(shape))
    ku = _u(k)
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        out = mix(mix(ku)[..., None] ^ (idx * M2))
    return _t(out.reshape(ku.shape + shape))


def randint(k, shape, minval: int, maxval: int) -> torch.Tensor:
    """Integers in [minval, maxval): the top 63 bits modulo the span."""
    span = int(maxval) - int(minval)
    b = _u(bits(k, shape)) >> np.uint64(1)
    if span <= 0:
        return torch.full(tuple(b.shape), int(minval), dtype=torch.int64)
    return torch.from_numpy((b % np.uint64(span)).astype(np.int64) + int(minval))


def uniform(k, shape=()) -> torch.Tensor:
    """U[0, 1) from the top 24 bits (exact in fp32)."""
    b = (_u(bits(k, shape)) >> np.uint64(40)).astype(np.float32)
    return torch.from_numpy(b * np.float32(1.0 / (1 << 24)))


def unit_uniform(k, shape=(), fold=None) -> torch.Tensor:
    """Zero-mean unit-variance uniform noise (u - 1/2) sqrt(12), optionally from
    fold_key(k, fold) — what the synthetic envs draw for observations."""
    if fold is not None:
        k = fold_key(k, fold)
    u = uniform(k, shape).numpy()
    return torch.from_numpy((u - np.float32(0.5)) * np.float32(3.4641016151377544))


def unit_normal(k, shape=(), fold=None) -> torch.Tensor:
    """N(0, 1) by Box-Muller from the two 24-bit halves of each hash (the optional law of the
    synthetic env, `mock_env.py:43,53`'s `jax.random.normal`; numpy float32 arithmetic)."""
    if fold is not None:
        k = fold_key(k, fold)
    b = _u(bits(k, shape))
    u1 = ((b >> np.uint64(40)) + np.uint64(1)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    u2 = ((b >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float32) \
        * np.float32(1.0 / (1 << 24))
    z = np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)
    return torch.as_tensor(z.astype(np.float32))


def permutation(k, n: int) -> torch.Tensor:
    """`jax.random.permutation(key, n)`: stable argsort of n 64-bit hashes, compared as
    SIGNED integers (the product sorts int64)."""
    h = _u(bits(k, (n,))).view(np.int64)
    return torch.from_numpy(np.argsort(h, kind="stable").astype(np.int64))


def permutations(k, n_perm: int, n: int) -> torch.Tensor:
    """stack([permutation(fold_in(k, e), n) for e in range(n_perm)])."""
    return torch.stack([permutation(fold_in(k, e), n) for e in range(n_perm)], dim=0)
