"""Minimal pytree utilities with JAX's conventions (the reference routes
observations, actions, rewards, carry state and rollout extras as PyTrees:
`nnx_ppo/algorithms/ppo.py:297-300,440-458`, `rollout.py:270-279`).

Conventions kept from `jax.tree`:
  * `None` is an empty subtree (no leaves), not a leaf;
  * dict leaves are visited in sorted-key order (so `Flattener` concatenates
    features in the same order as `jax.tree.flatten`, `networks/utils.py:104-108`);
  * dataclasses deriving from `TreeDataclass` are nodes whose children are their
    fields in declaration order (`nnx_ppo/jax_dataclass.py:17-41`).
"""
from __future__ import annotations

import dataclasses
from collections.abc import Mapping
from typing import Any, Callable


class TreeDataclass:
    """Frozen-dataclass mixin: pytree node with `.replace(**kw)`
    (counterpart of `nnx_ppo/jax_dataclass.py:17-41`)."""

    def replace(self, **kwargs):
        return dataclasses.replace(self, **kwargs)


def _is_node(x: Any) -> bool:
    return x is None or isinstance(x, (Mapping, list, tuple, TreeDataclass))


def _sorted_keys(d: Mapping):
    try:
        return sorted(d.keys())
    except TypeError:
        return list(d.keys())


def tree_leaves(tree: Any, is_leaf: Callable[[Any], bool] | None = None) -> list:
    out: list = []

    def rec(x):
        if is_leaf is not None and is_leaf(x):
            out.append(x)
        elif x is None:
            return
        elif isinstance(x, Mapping):
            for k in _sorted_keys(x):
                rec(x[k])
        elif isinstance(x, (list, tuple)):
            for v in x:
                rec(v)
        elif isinstance(x, TreeDataclass):
            for f in dataclasses.fields(x):
                rec(getattr(x, f.name))
        else:
            out.append(x)

    rec(tree)
    return out


def tree_map(fn: Callable, tree: Any, *rest: Any,
             is_leaf: Callable[[Any], bool] | None = None) -> Any:
    """Map `fn` over the leaves of `tree`; `rest` must share its structure."""

    def rec(x, others):
        if is_leaf is not None and is_leaf(x):
            return fn(x, *others)
        if x is None:
            return None
        if isinstance(x, Mapping):
            for o in others:
                if not isinstance(o, Mapping) or set(o.keys()) != set(x.keys()):
                    raise ValueError("tree_map: dict structure mismatch")
            return {k: rec(x[k], [o[k] for o in others]) for k in x.keys()}
        if isinstance(x, (list, tuple)):
            for o in others:
                if not isinstance(o, (list, tuple)) or len(o) != len(x):
                    raise ValueError("tree_map: sequence structure mismatch")
            vals = [rec(v, [o[i] for o in others]) for i, v in enumerate(x)]
            if isinstance(x, tuple) and hasattr(x, "_fields"):  # namedtuple
                return type(x)(*vals)
            return type(x)(vals)
        if isinstance(x, TreeDataclass):
            kw = {}
            for f in dataclasses.fields(x):
                kw[f.name] = rec(getattr(x, f.name), [getattr(o, f.name) for o in others])
            obj = object.__new__(type(x))
            for k, v in kw.items():
                object.__setattr__(obj, k, v)
            return obj
        return fn(x, *others)

    return rec(tree, list(rest))


def tree_reduce(fn: Callable, tree: Any, initializer=None):
    leaves = tree_leaves(tree)
    if initializer is None:
        if not leaves:
            raise ValueError("tree_reduce of an empty tree with no initializer")
        acc, leaves = leaves[0], leaves[1:]
    else:
        acc = initializer
    for x in leaves:
        acc = fn(acc, x)
    return acc


def tree_all(tree: Any) -> bool:
    return all(bool(x) for x in tree_leaves(tree))


def canonicalize(obj: Any) -> Any:
    """Mappings → plain dict, recursively (`networks/normalizer.py:18-32`)."""
    if isinstance(obj, Mapping):
        return {k: canonicalize(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [canonicalize(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(canonicalize(v) for v in obj)
    return obj
