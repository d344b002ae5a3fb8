"""Thin Python wrappers over the libmippo C ABI (one per entry point).

Each wrapper validates shapes/dtypes on the host (so a kernel never sees an
operand it was not sized for), allocates the outputs with torch, and enqueues
the kernel on torch's current stream.  No wrapper synchronises.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import MippoError, check, lib, ptr, stream

f32 = torch.float32
u8 = torch.uint8


def _as_u8(t: torch.Tensor) -> torch.Tensor:
    """bool tensors share storage layout with uint8 (0/1)."""
    if t.dtype == torch.bool:
        return t.view(torch.uint8)
    if t.dtype != torch.uint8:
        raise MippoError(f"flag tensor must be bool or uint8, got {t.dtype}")
    return t


def gae(
    rewards: torch.Tensor,
    values: torch.Tensor,
    last_value: torch.Tensor,
    done: torch.Tensor,
    truncated: torch.Tensor,
    gamma: float,
    lambda_: float,
    with_targets: bool = False,
    out: torch.Tensor | None = None,
    out_targets: torch.Tensor | None = None,
):
    """GAE reverse scan over `[T, N]` (reference ppo.py:351-394).

    Returns `advantages` or `(advantages, targets)` with `targets = V + A`.
    """
    if rewards.dim() != 2:
        raise MippoError(f"rewards must be [T, N], got {tuple(rewards.shape)}")
    T, N = rewards.shape
    if values.shape != (T, N) or done.shape != (T, N) or truncated.shape != (T, N):
        raise MippoError("gae: values/done/truncated must match rewards [T, N]")
    if last_value.shape != (N,):
        raise MippoError(f"gae: last_value must be [{N}], got {tuple(last_value.shape)}")
    done = _as_u8(done)
    truncated = _as_u8(truncated)
    adv = out if out is not None else torch.empty_like(rewards)
    tgt = None
    if with_targets:
        tgt = out_targets if out_targets is not None else torch.empty_like(rewards)
    rc = lib().mi_gae_f32(
        ptr(rewards, f32), ptr(values, f32), ptr(last_value, f32),
        ptr(done, u8), ptr(truncated, u8), ptr(adv, f32), ptr(tgt, f32),
        T, N, float(gamma), float(lambda_), stream(),
    )
    check(rc, "mi_gae_f32")
    return (adv, tgt) if with_targets else adv
