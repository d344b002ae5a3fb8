"""Weight initialisers in the role of `nnx.initializers.*` (third-party flax
code, not in the reference tree: published formulas restated, PARITY UNPINNED).
An initialiser is `f(generator, shape) -> np.ndarray` with shape = (fan_in, fan_out)."""
from __future__ import annotations

import numpy as np


def variance_scaling(scale: float, mode: str, distribution: str):
    """flax/jax `variance_scaling`: variance = scale / fan ('fan_in' | 'fan_out' |
    'fan_avg'); 'uniform' -> U(±sqrt(3 var)); 'normal'; 'truncated_normal' (±2σ,
    std corrected by 0.87962566103423978).  Used by `factories.py:112-114`."""

    def init(gen: np.random.Generator, shape):
        fan_in, fan_out = shape[0], shape[-1]
        fan = {"fan_in": fan_in, "fan_out": fan_out, "fan_avg": (fan_in + fan_out) / 2}[mode]
        var = scale / max(1.0, fan)
        if distribution == "uniform":
            lim = np.sqrt(3.0 * var)
            return gen.uniform(-lim, lim, size=shape).astype(np.float32)
        if distribution == "normal":
            return (gen.standard_normal(size=shape) * np.sqrt(var)).astype(np.float32)
        if distribution == "truncated_normal":
            std = np.sqrt(var) / 0.87962566103423978
            x = gen.standard_normal(size=shape)
            bad = np.abs(x) > 2.0
            while bad.any():
                x[bad] = gen.standard_normal(size=int(bad.sum()))
                bad = np.abs(x) > 2.0
            return (x * std).astype(np.float32)
        raise ValueError(distribution)

    return init


def lecun_normal():
    """flax.nnx.Linear's default kernel_init."""
    return variance_scaling(1.0, "fan_in", "truncated_normal")


def orthogonal(scale: float = 1.0):
    """flax/jax `orthogonal`: Q of the QR decomposition of a Gaussian matrix, columns
    sign-fixed by diag(R) — flax.nnx.LSTMCell's default recurrent_kernel_init."""

    def init(gen: np.random.Generator, shape):
        rows, cols = shape[0], shape[-1]
        a = gen.standard_normal(size=(max(rows, cols), min(rows, cols)))
        q, r = np.linalg.qr(a)
        q = q * np.sign(np.diag(r))
        if rows < cols:
            q = q.T
        return (scale * q[:rows, :cols]).astype(np.float32)

    return init


def zeros(gen, shape):
    return np.zeros(shape, dtype=np.float32)
