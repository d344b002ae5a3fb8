"""PPOAdapter: two-port router from network output to `PPONetworkOutput`
(counterpart of `nnx_ppo/networks/adapter.py:48-133`).  Routing only — both
ports see the same input; the action port's output is a tree of sampler dicts
`{"action", "log_likelihood"}`; the value port's trailing singleton axis is
squeezed (`adapter.py:55-58,98`)."""
from __future__ import annotations

import os

from typing import Any

import torch

from ..tree import tree_leaves, tree_map
from .types import (ModuleState, PPONetworkOutput, StatefulModule, StatefulModuleOutput,
                    add_reg)

_SAMPLER_DICT_KEYS = frozenset({"action", "log_likelihood"})


def _is_sampler_dict(x: Any) -> bool:
    return isinstance(x, dict) and _SAMPLER_DICT_KEYS.issubset(x.keys())


def _squeeze_trailing_one(v: Any) -> Any:
    if hasattr(v, "shape") and len(v.shape) and v.shape[-1] == 1:
        return v.squeeze(-1)
    return v


class _Fork:
    """Run the value port on a second HIP stream while the action port runs on the
    current one.  The two ports are independent (both read the same input,
    adapter.py:75-117), and at this workload's sizes neither fills the chip, so
    overlapping them shortens the critical path; inside a captured HIP graph the
    fork/join become parallel branches.  Tensors that cross streams are
    registered with the caching allocator (`record_stream`)."""

    def __init__(self, *inputs):
        self.main = torch.cuda.current_stream()
        dev = self.main.device
        side = _SIDE_STREAMS.get(dev)
        if side is None:
            side = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
        self.side = side
        self.inputs = [t for x in inputs for t in tree_leaves(x) if isinstance(t, torch.Tensor)]

    def __enter__(self):
        self.side.wait_stream(self.main)
        for t in self.inputs:
            t.record_stream(self.side)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        return False

    def join(self, *outputs):
        self.main.wait_stream(self.side)
        for x in outputs:
            for t in tree_leaves(x):
                if isinstance(t, torch.Tensor):
                    t.record_stream(self.main)


_SIDE_STREAMS: dict = {}
# MIPPO_OVERLAP_PORTS = 1 / 0 forces the second stream on / off; unset, a PPOAdapter
# overlaps its ports only when they are WIDE (a Dense wider than 256: per-layer GEMMs of
# 100+ us).  A second branch in a captured graph costs a few us of submission per node,
# more than it hides for narrow ports (measured, one box: C4, GRU(64) actor / 2x256
# critic: 12.4 M env-steps/s with the fork, 13.9 M without; C3, 4x256 / 2x512: 17.7 M
# with, 17.2 M without).
_OVERLAP_ENV = os.environ.get("MIPPO_OVERLAP_PORTS")
OVERLAP_PORTS = _OVERLAP_ENV != "0"


def _can_fork(x, wide: bool = True) -> bool:
    leaves = tree_leaves(x)
    if not (OVERLAP_PORTS and (wide or _OVERLAP_ENV == "1")):
        return False
    return bool(leaves) and isinstance(leaves[0], torch.Tensor) and leaves[0].is_cuda


def _is_wide(*ports) -> bool:
    for port in ports:
        for m in port.modules():
            if max(getattr(m, "in_features", 0), getattr(m, "out_features", 0)) > 256:
                return True
    return False


class PPOAdapter(StatefulModule):
    def __init__(self, action: StatefulModule, value: StatefulModule):
        self.action = action
        self.value = value
        self._wide = _is_wide(action, value)

    def __call__(self, state: dict[str, ModuleState], x: Any,
                 rollout_extras: Any = None) -> StatefulModuleOutput:
        if rollout_extras is None:
            a_re = v_re = None
        else:
            a_re = rollout_extras["action"]
            v_re = rollout_extras["value"]
        if _can_fork(x, self._wide):
            fork = _Fork(x, state["value"], v_re)
            with fork:
                v_out = self.value(state["value"], x, v_re)
            a_out = self.action(state["action"], x, a_re)
            fork.join(v_out.output, v_out.next_state, v_out.rollout_extras,
                      v_out.regularization_loss)
        else:
            a_out = self.action(state["action"], x, a_re)
            v_out = self.value(state["value"], x, v_re)
        actions = tree_map(lambda d: d["action"], a_out.output, is_leaf=_is_sampler_dict)
        loglikelihoods = tree_map(lambda d: d["log_likelihood"], a_out.output,
                                  is_leaf=_is_sampler_dict)
        value_estimates = tree_map(_squeeze_trailing_one, v_out.output)
        return StatefulModuleOutput(
            next_state={"action": a_out.next_state, "value": v_out.next_state},
            output=PPONetworkOutput(actions=actions, loglikelihoods=loglikelihoods,
                                    value_estimates=value_estimates),
            regularization_loss=add_reg(a_out.regularization_loss, v_out.regularization_loss),
            metrics={"action": a_out.metrics, "value": v_out.metrics},
            rollout_extras={"action": a_out.rollout_extras, "value": v_out.rollout_extras},
        )

    def forward_value(self, state: dict[str, ModuleState], x: Any) -> Any:
        v_out = self.value(state["value"], x, None)
        return tree_map(_squeeze_trailing_one, v_out.output)

    def initialize_state(self, batch_size: int) -> dict[str, ModuleState]:
        return {"action": self.action.initialize_state(batch_size),
                "value": self.value.initialize_state(batch_size)}

    def reset_state(self, prev_state: dict[str, ModuleState]) -> dict[str, ModuleState]:
        return {"action": self.action.reset_state(prev_state["action"]),
                "value": self.value.reset_state(prev_state["value"])}

    def update_statistics(self, rollout_extras: Any) -> None:
        self.action.update_statistics(rollout_extras["action"])
        self.value.update_statistics(rollout_extras["value"])

    # ---- training protocol ------------------------------------------------------
    def replay(self, state0, x_seq, done_seq, extras_seq, need_input_grad=True, x_value=None):
        """`x_value` (containers.Sequential.replay_with_bootstrap): the value port's input when
        it is longer than the action port's — `[T + 1, B, ...]`, step T being the bootstrap
        observation, for a stateless row-wise value port; its output keeps the extra step and
        the caller splits it."""
        a_re = None if extras_seq is None else extras_seq["action"]
        v_re = None if extras_seq is None else extras_seq["value"]
        xv = x_seq if x_value is None else x_value
        if _can_fork(x_seq, self._wide):
            fork = _Fork(xv, state0["value"], v_re, done_seq)
            with fork:
                v_ctx, v_out, v_reg, v_fs = self.value.replay(state0["value"], xv, done_seq,
                                                              v_re, need_input_grad)
            a_ctx, a_out, a_reg, a_fs = self.action.replay(state0["action"], x_seq, done_seq,
                                                           a_re, need_input_grad)
            fork.join(v_out, v_reg, v_fs)
        else:
            a_ctx, a_out, a_reg, a_fs = self.action.replay(state0["action"], x_seq, done_seq,
                                                           a_re, need_input_grad)
            v_ctx, v_out, v_reg, v_fs = self.value.replay(state0["value"], xv, done_seq, v_re,
                                                          need_input_grad)
        actions = tree_map(lambda d: d["action"], a_out, is_leaf=_is_sampler_dict)
        loglik = tree_map(lambda d: d["log_likelihood"], a_out, is_leaf=_is_sampler_dict)
        values = tree_map(_squeeze_trailing_one, v_out)
        squeezed = tree_map(lambda v, s: v.shape != s.shape, v_out, values)
        out = PPONetworkOutput(actions=actions, loglikelihoods=loglik, value_estimates=values)
        ctx = (a_ctx, v_ctx, a_out, squeezed)
        return ctx, out, add_reg(a_reg, v_reg), {"action": a_fs, "value": v_fs}

    def replay_backward(self, ctx, g_out: PPONetworkOutput, g_reg):
        a_ctx, v_ctx, a_out, squeezed = ctx
        # gradient tree for the action port: sampler dicts with d/d log_likelihood
        if _is_sampler_dict(a_out):
            g_a = {"action": None, "log_likelihood": g_out.loglikelihoods}
        else:
            g_a = tree_map(lambda d, g: {"action": None, "log_likelihood": g}, a_out,
                           g_out.loglikelihoods, is_leaf=_is_sampler_dict)
        g_v = tree_map(lambda g, sq: g.unsqueeze(-1) if sq else g, g_out.value_estimates,
                       squeezed)
        if _can_fork(g_v, self._wide):
            fork = _Fork(g_v)
            with fork:
                gx_v = self.value.replay_backward(v_ctx, g_v, g_reg)
            gx_a = self.action.replay_backward(a_ctx, g_a, g_reg)
            fork.join(gx_v)
        else:
            gx_a = self.action.replay_backward(a_ctx, g_a, g_reg)
            gx_v = self.value.replay_backward(v_ctx, g_v, g_reg)
        if gx_a is None and gx_v is None:
            return None
        if gx_a is None:
            return gx_v
        if gx_v is None:
            return gx_a
        return tree_map(lambda a, b: a + b, gx_a, gx_v)
