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


def constant(shape, dtype, value, device) -> torch.Tensor:
    key = (tuple(shape), dtype, value, str(device))
    t = _cache.get(key)
    if t is None:
        t = torch.full(tuple(shape), value, dtype=dtype, device=device)
        _cache[key] = t
    return t
