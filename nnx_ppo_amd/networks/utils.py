"""Stateless PyTree plumbing (counterpart of `nnx_ppo/networks/utils.py`;
`Flattener` 65-116 is what a PyTree-observation network needs)."""
from __future__ import annotations

from typing import Any

import torch

from ..tree import tree_leaves
from .types import StatefulModule, StatefulModuleOutput, zero_scalar


def _flatten_at_depth(x: Any, preserve_levels: int, lead: int) -> Any:
    """utils.py:100-116.  `lead` = number of leading batch axes kept (1 for a
    single step `[B, ...]`, 2 for a sequence `[T, B, ...]`)."""
    if preserve_levels == 0:
        leaves = tree_leaves(x)
        return torch.cat([a.reshape(*a.shape[:lead], -1) for a in leaves], dim=-1)
    if isinstance(x, dict):
        return {k: _flatten_at_depth(v, preserve_levels - 1, lead) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_flatten_at_depth(v, preserve_levels - 1, lead) for v in x)
    raise TypeError(
        "Flattener(preserve_levels > 0) requires dict/list/tuple at each preserved level; "
        f"encountered a leaf of type {type(x).__name__} with {preserve_levels} levels still "
        "to preserve.")


class Flattener(StatefulModule):
    """Flatten a pytree into one `[B, F]` tensor (leaves in sorted-key order, as
    `jax.tree.flatten`), or keep the top `preserve_levels` levels of structure."""

    def __init__(self, preserve_levels: int = 0):
        if preserve_levels < 0:
            raise ValueError(f"preserve_levels must be >= 0, got {preserve_levels}")
        self.preserve_levels = preserve_levels

    def __call__(self, state, x: Any, rollout_extras: Any = None) -> StatefulModuleOutput:
        out = _flatten_at_depth(x, self.preserve_levels, 1)
        dev = tree_leaves(x)[0].device
        return StatefulModuleOutput((), out, zero_scalar(dev), {}, None)

    def replay(self, state0, x_seq, done_seq, extras_seq, need_input_grad=True):
        if need_input_grad:
            raise NotImplementedError(
                "Flattener.replay_backward: gradients w.r.t. a flattened PyTree input are "
                "not needed by any supported network (no trainable layer upstream)")
        return None, _flatten_at_depth(x_seq, self.preserve_levels, 2), None, ()

    def replay_backward(self, ctx, g_out, g_reg):
        return None
