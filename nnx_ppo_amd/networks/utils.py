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


class Filter(StatefulModule):
    """utils.py:119-165 — declarative pytree extraction: `spec = {output_key:
    extraction}` with extraction a key (`x[k]`), a tuple path (`x[k1][k2]...`) or a
    callable applied to the whole input.  Everything not named is dropped."""

    def __init__(self, spec: dict):
        if not isinstance(spec, dict):
            raise TypeError(f"Filter spec must be a dict; got {type(spec).__name__}")
        for out_key, sub in spec.items():
            if not isinstance(sub, (str, tuple)) and not callable(sub):
                raise TypeError(f"Filter spec for {out_key!r} must be str, tuple, or callable; "
                                f"got {type(sub).__name__}")
        self._spec = dict(spec)

    def _extract(self, x):
        output = {}
        for out_key, sub in self._spec.items():
            if isinstance(sub, str):
                output[out_key] = x[sub]
            elif isinstance(sub, tuple):
                v = x
                for p in sub:
                    v = v[p]
                output[out_key] = v
            else:
                output[out_key] = sub(x)
        return output

    def __call__(self, state, x: Any, rollout_extras: Any = None) -> StatefulModuleOutput:
        dev = tree_leaves(x)[0].device
        return StatefulModuleOutput((), self._extract(x), zero_scalar(dev), {}, None)

    def replay(self, state0, x_seq, done_seq, extras_seq, need_input_grad=True):
        if need_input_grad:
            raise NotImplementedError(
                "Filter.replay_backward: gradients w.r.t. a filtered PyTree input are not "
                "needed by any supported network (no trainable layer upstream)")
        return None, self._extract(x_seq), None, ()

    def replay_backward(self, ctx, g_out, g_reg):
        return None


class Scale(StatefulModule):
    """utils.py:168-183 — multiply every leaf by a fixed scalar."""

    def __init__(self, factor: float):
        self.factor = float(factor)

    def _scale(self, x):
        from ..tree import tree_map

        return tree_map(lambda v: v * self.factor, x)

    def __call__(self, state, x: Any, rollout_extras: Any = None) -> StatefulModuleOutput:
        dev = tree_leaves(x)[0].device
        return StatefulModuleOutput(state, self._scale(x), zero_scalar(dev), {}, None)

    def replay(self, state0, x_seq, done_seq, extras_seq, need_input_grad=True):
        return need_input_grad, self._scale(x_seq), None, state0

    def replay_backward(self, ctx, g_out, g_reg):
        return self._scale(g_out) if ctx else None


def _keyed_base():
    from .containers import _Keyed

    return _Keyed


class Merge(_keyed_base()):
    """utils.py:186-255 — named sub-modules on the SAME input, each returning a dict;
    the dicts are merged into one flat dict, duplicate keys are an error."""

    _KIND = "Merge"

    def _combine(self, outputs: dict):
        merged: dict = {}
        self._owner: dict = {}
        for name, out in outputs.items():
            if not isinstance(out, dict):
                raise TypeError(f"Merge component {name!r} must return a dict; got "
                                f"{type(out).__name__}")
            for k, v in out.items():
                if k in merged:
                    raise ValueError(f"Merge: duplicate key {k!r} produced by multiple components")
                merged[k] = v
                self._owner[k] = name
        return merged

    def _split_grad(self, g_out, meta: dict) -> dict:
        grads: dict = {name: {} for name in meta}
        for k, g in g_out.items():
            grads[self._owner[k]][k] = g
        return grads


class Map(_keyed_base()):
    """utils.py:258-326 — per-key dispatch: dict input, dict output; extra input keys
    are dropped."""

    _KIND = "Map"
    _PER_KEY_INPUT = True

    def _combine(self, outputs: dict):
        return outputs

    def _split_grad(self, g_out, meta: dict) -> dict:
        return {k: g_out[k] for k in meta}
