"""Cached constant tensors for env code.

A reset state is mostly constants (zero step counters, zero rewards, false
flags) and the rollout computes a reset state for ALL envs every step
(`rollout.py:39`), so filling them would cost a launch each, 30 times per
iteration.  `constant(shape, dtype, value, device)` returns a shared READ-ONLY
tensor instead.  Nothing may write to it: the rollout only ever reads reset
states (as `tree_where` operands), and `new_training_state` clones the initial
env state before it becomes a static HIP-graph buffer."""
from __future__ import annotations

import torch

_cache: dict = {}
_ids: dict = {}
_zero_ids: dict = {}


def constant(shape, dtype, value, device) -> torch.Tensor:
    key = (tuple(shape), dtype, value, str(device))
    t = _cache.get(key)
    if t is None:
        t = torch.full(tuple(shape), value, dtype=dtype, device=device)
        _cache[key] = t
        _ids[id(t)] = t
        if value == 0:
            _zero_ids[id(t)] = t
    return t


def is_zero_constant(t) -> bool:
    """True for an all-zero tensor handed out by `constant` (identity, not value)."""
    return isinstance(t, torch.Tensor) and _zero_ids.get(id(t)) is t


def is_constant(t) -> bool:
    """True for a tensor handed out by `constant` (identity, not value)."""
    return isinstance(t, torch.Tensor) and _ids.get(id(t)) is t


def cast_constant(t: torch.Tensor, dtype) -> torch.Tensor:
    """`t.to(dtype)` of a cached constant, itself cached."""
    key = ("cast", id(t), dtype)
    c = _cache.get(key)
    if c is None:
        c = t.to(dtype)
        _cache[key] = c
        _ids[id(c)] = c
    return c
