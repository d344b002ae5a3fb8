"""Philox4x32-10 + Box-Muller in numpy.  TEST INFRASTRUCTURE — see oracle/__init__.py.

Independent restatement of the published algorithm (Salmon, Moraes, Dror, Shaw:
"Parallel random numbers: as easy as 1, 2, 3", SC'11; constants M0=0xD2511F53,
M1=0xCD9E8D57, W0=0x9E3779B9, W1=0xBB67AE85) and of the noise scheme the HIP
sampler kernels use (nnx_ppo_amd/csrc/philox.h):

    counter = (elem_lo, elem_hi, offset_lo, offset_hi), key = (seed_lo, seed_hi)
    eps = BoxMuller(x0, x1), eps2 = BoxMuller(x2, x3)
    u1 = ((a >> 8) + 0.5) / 2^24, u2 = (b >> 8) / 2^24
    BoxMuller = sqrt(-2 ln u1) * cos(2 pi u2)

The reference's own noise (jax.random threefry via nnx.Rngs,
nnx_ppo/networks/sampling_layers.py:96,144) cannot be reproduced without JAX —
parity on noise VALUES is unpinned; what is checked is that device and oracle
agree on this scheme (integers exactly, normals to fp32 rounding).
Known-answer vectors from the Random123 distribution pin the block function.
"""
from __future__ import annotations

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Vectorised block function: c* are uint32 arrays (broadcastable), k* ints."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64)
    c3 = np.asarray(c3, dtype=np.uint64)
    k0 &= 0xFFFFFFFF
    k1 &= 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & _MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & _MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32),
            c3.astype(np.uint32))


def box_muller(a, b):
    u1 = ((a >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0
    u2 = (b >> np.uint32(8)).astype(np.float64) / 16777216.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def normal_pair(seed: int, offset: int, n: int):
    """(eps, eps2) float64 arrays of length n for elements 0..n-1 of call `offset`."""
    seed &= 0xFFFFFFFFFFFFFFFF
    offset &= 0xFFFFFFFFFFFFFFFF
    e = np.arange(n, dtype=np.uint64)
    x0, x1, x2, x3 = philox4x32_10(
        e & _MASK, e >> np.uint64(32),
        np.full(n, offset & 0xFFFFFFFF, dtype=np.uint64),
        np.full(n, offset >> 32, dtype=np.uint64),
        seed & 0xFFFFFFFF, seed >> 32)
    return box_muller(x0, x1), box_muller(x2, x3)
