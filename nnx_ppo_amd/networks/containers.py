"""State / extras routing containers (counterpart of
`nnx_ppo/networks/containers.py`; `Sequential` 14-52 is on the hot path)."""
from __future__ import annotations

from collections.abc import Sequence
from typing import Any

from .. import config
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
        runs = self._dense_runs() if config.compute_dtype() == "bf16" else {}
        i = 0
        n = len(self.layers)
        while i < n:
            if i in runs:
                # a run of Dense layers: one bf16 chain (state / extras / metrics of a
                # Dense are (), None, {} — feedforward.py:51)
                from . import dense_chain

                j = runs[i]
                lead = x.shape[:-1]
                y = dense_chain.forward_infer(self.layers[i:j], x.reshape(-1, x.shape[-1]))
                x = y.view(*lead, y.shape[-1])
                for k in range(i, j):
                    new_state.append(network_state[k])
                    new_extras.append(None)
                    metrics[len(metrics)] = {}
                i = j
                continue
            layer, layer_state = self.layers[i], network_state[i]
            layer_extras = None if rollout_extras is None else rollout_extras[i]
            out = layer(layer_state, x, layer_extras)
            new_state.append(out.next_state)
            new_extras.append(out.rollout_extras)
            x = out.output
            reg = add_reg(reg, out.regularization_loss)
            metrics[len(metrics)] = out.metrics
            i += 1
        return StatefulModuleOutput(new_state, x, reg, metrics, new_extras)

    def _dense_runs(self) -> dict:
        """start index -> end index (exclusive) of every maximal run of Dense layers."""
        from .feedforward import Dense

        runs, i, n = {}, 0, len(self.layers)
        while i < n:
            if type(self.layers[i]) is Dense:
                j = i
                while j < n and type(self.layers[j]) is Dense:
                    j += 1
                runs[i] = j
                i = j
            else:
                i += 1
        return runs

    def forward_value(self, network_state: list[ModuleState], obs: Any) -> Any:
        x = obs
        for layer, layer_state in zip(self.layers[:-1], network_state[:-1]):
            x = layer(layer_state, x, None).output
        return self.layers[-1].forward_value(network_state[-1], x)

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
        runs = self._dense_runs() if config.compute_dtype() == "bf16" else {}
        i, n = 0, len(self.layers)
        while i < n:
            if i in runs:
                from . import dense_chain

                j = runs[i]
                lead = x.shape[:-1]
                cctx, y = dense_chain.forward_train(self.layers[i:j],
                                                    x.reshape(-1, x.shape[-1]), upstream_needs)
                x = y.view(*lead, y.shape[-1])
                ctxs.append(("chain", i, j, cctx, lead))
                final_state.extend(state0[i:j])
                upstream_needs = True
                i = j
                continue
            layer = self.layers[i]
            layer_extras = None if extras_seq is None else extras_seq[i]
            ctx, x, r, fs = layer.replay(state0[i], x, done_seq, layer_extras,
                                         need_input_grad=upstream_needs)
            ctxs.append(("layer", i, ctx))
            final_state.append(fs)
            reg = add_reg(reg, r)
            upstream_needs = upstream_needs or bool(layer.parameters())
            i += 1
        return ctxs, x, reg, final_state

    def replay_backward(self, ctxs, g_out, g_reg):
        g = g_out
        for entry in reversed(ctxs):
            if entry[0] == "chain":
                from . import dense_chain

                _, i, j, cctx, lead = entry
                g2 = g.reshape(-1, g.shape[-1])
                if not g2.is_contiguous():
                    g2 = g2.contiguous()
                gi = dense_chain.backward(self.layers[i:j], cctx, g2)
                g = None if gi is None else gi.view(*lead, gi.shape[-1])
            else:
                _, i, ctx = entry
                g = self.layers[i].replay_backward(ctx, g, g_reg)
            if g is None:
                return None
        return g
