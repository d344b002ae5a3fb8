"""Running observation normaliser (counterpart of
`nnx_ppo/networks/normalizer.py:35-136`).  Forward reads the statistics and
never writes them (contract of `docs/reference/contexts.rst:82-95`); the raw
input is emitted as `rollout_extras` and folded in once per training step by
`update_statistics` (batched Welford merge, `normalizer.py:98-136`).  A PyTree
`shape` gives per-leaf statistics sharing one counter (`normalizer.py:52-61`)."""
from __future__ import annotations

from typing import Any

import torch

from .. import ops, parallel
from ..tree import canonicalize, tree_leaves, tree_map
from .types import StatefulModule, StatefulModuleOutput, Variable, zero_scalar


def _zeros(shape):
    if isinstance(shape, int):
        shape = (shape,)
    return torch.zeros(tuple(shape), dtype=torch.float32)


def _is_shape(x) -> bool:
    return isinstance(x, int) or (
        isinstance(x, (tuple, list)) and all(isinstance(v, int) for v in x))


class Normalizer(StatefulModule):
    def __init__(self, shape):
        if _is_shape(shape):
            self.mean = Variable(_zeros(shape))
            self.M2 = Variable(_zeros(shape))
        else:
            shape = canonicalize(shape)
            self.mean = Variable(tree_map(_zeros, shape, is_leaf=_is_shape))
            self.M2 = Variable(tree_map(_zeros, shape, is_leaf=_is_shape))
        self.counter = Variable(torch.zeros(1, dtype=torch.float32))
        self.epsilon = 1e-6

    def _normalize(self, x):
        cnt = self.counter.value
        return tree_map(
            lambda v, m, m2: ops.normalize_fwd(v if v.is_contiguous() else v.contiguous(),
                                               m, m2, cnt, self.epsilon),
            x, self.mean.value, self.M2.value)

    def __call__(self, state, x: Any, rollout_extras: Any = None) -> StatefulModuleOutput:
        x = canonicalize(x)
        out = self._normalize(x)
        dev = tree_leaves(x)[0].device
        return StatefulModuleOutput(next_state=(), output=out,
                                    regularization_loss=zero_scalar(dev), metrics={},
                                    rollout_extras=x)

    def update_statistics(self, rollout_extras: Any) -> None:
        """normalizer.py:98-136 — `rollout_extras` leaves are `[T, B, *feat]`."""
        xs = tree_leaves(canonicalize(rollout_extras))
        means = tree_leaves(self.mean.value)
        m2s = tree_leaves(self.M2.value)
        assert len(xs) == len(means) == len(m2s)
        cnt = self.counter.value
        for i, (x, mean, m2) in enumerate(zip(xs, means, m2s)):
            F = mean.numel()
            stats = ops.welford_batch_stats(x if x.is_contiguous() else x.contiguous(), F)
            stats = parallel.merge_batch_stats(stats)
            ops.welford_merge(mean.view(-1), m2.view(-1), cnt, stats,
                              advance_counter=(i == len(xs) - 1))

    # ---- training protocol -------------------------------------------------------
    def replay(self, state0, x_seq, done_seq, extras_seq, need_input_grad=True):
        x_seq = canonicalize(x_seq)
        return need_input_grad, self._normalize(x_seq), None, ()

    def replay_backward(self, ctx, g_out, g_reg):
        if not ctx:
            return None
        cnt = self.counter.value
        return tree_map(lambda g, m2: ops.normalize_bwd(g.contiguous(), m2, cnt, self.epsilon),
                        g_out, self.M2.value)
