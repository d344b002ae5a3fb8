"""State / extras routing containers (counterpart of
`nnx_ppo/networks/containers.py`; `Sequential` 14-52 is on the hot path)."""
from __future__ import annotations

from collections.abc import Sequence
from typing import Any

from .types import ModuleState, StatefulModule, StatefulModuleOutput, add_reg, zero_scalar


def _device_of(x):
    from ..tree import tree_leaves

    leaves = tree_leaves(x)
    return leaves[0].device if leaves else "cpu"


class Sequential(StatefulModule):
    """containers.py:14-52 — chain layers; state, extras and metrics are lists /
    index-keyed dicts parallel to the layer list; regularisers are summed."""

    def __init__(self, layers: Sequence[StatefulModule]):
        self.layers = list(layers)

    def __call__(self, network_state: list[ModuleState], obs: Any,
                 rollout_extras: Any = None) -> StatefulModuleOutput:
        new_state = []
        new_extras: list[Any] = []
        x = obs
        reg = zero_scalar(_device_of(obs))
        metrics = {}
        for i, (layer, layer_state) in enumerate(zip(self.layers, network_state)):
            layer_extras = None if rollout_extras is None else rollout_extras[i]
            out = layer(layer_state, x, layer_extras)
            new_state.append(out.next_state)
            new_extras.append(out.rollout_extras)
            x = out.output
            reg = add_reg(reg, out.regularization_loss)
            metrics[len(metrics)] = out.metrics
        return StatefulModuleOutput(new_state, x, reg, metrics, new_extras)

    def initialize_state(self, batch_size: int) -> list[ModuleState]:
        return [layer.initialize_state(batch_size) for layer in self.layers]

    def reset_state(self, prev_state: list[ModuleState]) -> list[ModuleState]:
        return [layer.reset_state(s) for layer, s in zip(self.layers, prev_state)]

    def update_statistics(self, rollout_extras: Any) -> None:
        for layer, layer_extras in zip(self.layers, rollout_extras):
            layer.update_statistics(layer_extras)

    def __getitem__(self, ind: int) -> StatefulModule:
        return self.layers[ind]

    def __len__(self) -> int:
        return len(self.layers)

    # ---- training protocol: layer by layer over the whole sequence ---------------
    def replay(self, state0, x_seq, done_seq, extras_seq, need_input_grad=True):
        ctxs = []
        final_state = []
        x = x_seq
        reg = None
        upstream_needs = need_input_grad
        for i, layer in enumerate(self.layers):
            layer_extras = None if extras_seq is None else extras_seq[i]
            ctx, x, r, fs = layer.replay(state0[i], x, done_seq, layer_extras,
                                         need_input_grad=upstream_needs)
            ctxs.append(ctx)
            final_state.append(fs)
            reg = add_reg(reg, r)
            upstream_needs = upstream_needs or bool(layer.parameters())
        return ctxs, x, reg, final_state

    def replay_backward(self, ctxs, g_out, g_reg):
        g = g_out
        for layer, ctx in zip(reversed(self.layers), reversed(ctxs)):
            g = layer.replay_backward(ctx, g, g_reg)
            if g is None:
                return None
        return g
