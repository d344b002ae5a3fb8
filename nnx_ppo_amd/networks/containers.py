"""State / extras routing containers (counterpart of
`nnx_ppo/networks/containers.py`; `Sequential` 14-52 is on the hot path)."""
from __future__ import annotations

import os
from collections.abc import Sequence
from typing import Any

import torch

from .. import config, ops
from .types import ModuleState, StatefulModule, StatefulModuleOutput, add_reg, zero_scalar


def _device_of(x):
    from ..tree import tree_leaves

    leaves = tree_leaves(x)
    return leaves[0].device if leaves else "cpu"


# the bootstrap observation as step T of a stateless value chain's replay
# (Sequential.replay_with_bootstrap); MIPPO_BOOTSTRAP_IN_CHAIN=0: a forward_value launch
BOOTSTRAP_IN_CHAIN = os.environ.get("MIPPO_BOOTSTRAP_IN_CHAIN", "1") != "0"
# a linear head + sampler behind a recurrent layer inside its sequence launch
# (recurrent.GRU.replay(tail=...)); MIPPO_REC_TAIL=0: their own launches (A/B, bit-identity tests)
REC_TAIL = os.environ.get("MIPPO_REC_TAIL", "1") != "0"
# ... and their backward in front of the BPTT inside ITS launch (GRU.replay_backward_tail);
# MIPPO_REC_TAIL_BWD=0: mi_tanh_gauss_bwd_f32 + mi_mlp_bwd_dx_bf16 + mi_gru_seq_bwd_bf16
REC_TAIL_BWD = os.environ.get("MIPPO_REC_TAIL_BWD", "1") != "0"
# ... and the recurrent layer's INPUT projection inside the sequence launches as well, forward
# and backward (GRU.replay(proj=...): no fp32 gi / dgi round trip, the Dense chain in front
# stops one layer earlier and its backward launch goes); MIPPO_REC_PROJ=0: the projection as
# the chain's last layer
REC_PROJ = os.environ.get("MIPPO_REC_PROJ", "1") != "0"
# ... and the relu Dense in front of that projection (at most 8 inputs: the first layer of
# make_gru_actor_critic's actor) inside the forward sequence launch too — the actor's loss replay
# is then TWO launches + its share of the dW launch; MIPPO_REC_FRONT=0: its own chain launch
REC_FRONT = os.environ.get("MIPPO_REC_FRONT", "1") != "0"


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
                x2 = x.reshape(-1, x.shape[-1])
                y = dense_chain.forward_infer(self.layers[i:j],
                                              x2 if x2.is_contiguous() else x2.contiguous())
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
                # a recurrent layer right behind the run: its input projection rides in the
                # run's launch as one more (linear) layer
                rec = self.layers[j] if j < n else None
                proj = getattr(rec, "chain_projection", lambda: None)() if rec is not None else None
                if proj is not None and x.dim() == 3:
                    lead = x.shape[:-1]
                    x2 = x.reshape(-1, x.shape[-1])
                    x2 = x2 if x2.is_contiguous() else x2.contiguous()
                    layer_extras = None if extras_seq is None else extras_seq[j]
                    # ... and a linear head + tanh-Gaussian sampler right behind the recurrent
                    # layer (make_gru_actor_critic's actor) ride in ITS launch: two launches
                    # leave the critical path of the gradient step (csrc/gru_mfma.hip: GruTail)
                    tail = None
                    if (REC_TAIL and j + 3 == n and extras_seq is not None
                            and isinstance(extras_seq[j + 2], torch.Tensor)
                            and hasattr(rec, "replay_tail_supported")
                            and rec.replay_tail_supported(lead[0], self.layers[j + 1],
                                                          self.layers[j + 2])):
                        tail = (self.layers[j + 1], self.layers[j + 2], extras_seq[j + 2])
                    # ... and the projection itself inside the sequence launches when the run
                    # is ONE relu Dense of the GRU's width whose input needs no gradient
                    # (csrc/gru_mfma.hip: GruProj)
                    y_bf = None
                    if (REC_PROJ and REC_TAIL_BWD and tail is not None and j - i == 1
                            and not upstream_needs
                            and rec.replay_proj_supported(lead[0], lead[1], self.layers[i],
                                                          self.layers[j + 1])
                            and ops.gru_seq_bwd_tail_supported(
                                lead[0], rec.hidden_features, self.layers[j + 1].out_features)):
                        if REC_FRONT and x2.dtype == torch.float32 \
                                and rec.replay_front_supported(lead[0], lead[1], self.layers[i],
                                                               self.layers[j + 1]):
                            y_bf = "front"  # the relu layer inside the sequence launch too
                        else:
                            cctx, _ = dense_chain.forward_train([self.layers[i]], x2, False,
                                                                want_out=False)
                            y_bf = cctx[0][-1][1]  # the post-relu bf16 image
                            if y_bf is None or tuple(y_bf.shape) != (x2.shape[0],
                                                                     rec.hidden_features):
                                y_bf = None
                    if isinstance(y_bf, str):
                        res = rec.replay(state0[j], None, done_seq, layer_extras,
                                         need_input_grad=True, tail=tail,
                                         proj=(None, (lead[0], lead[1]), self.layers[i], x2))
                        cctx = res[4][4]
                    elif y_bf is not None:
                        res = rec.replay(state0[j], None, done_seq, layer_extras,
                                         need_input_grad=True, tail=tail,
                                         proj=(y_bf, (lead[0], lead[1])))
                    else:
                        chain = list(self.layers[i:j]) + [proj]
                        cctx, gi2 = dense_chain.forward_train(chain, x2, upstream_needs)
                        res = rec.replay(state0[j], None, done_seq, layer_extras,
                                         need_input_grad=True,
                                         gi_seq=gi2.view(*lead, gi2.shape[-1]),
                                         **({"tail": tail} if tail is not None else {}))
                    rctx, x, r, fs = res[:4]
                    ctxs.append(("chain+rec", i, j, cctx, lead, rctx))
                    final_state.extend(state0[i:j])
                    final_state.append(fs)
                    reg = add_reg(reg, r)
                    upstream_needs = True
                    i = j + 1
                    if tail is not None:
                        head_ctx, samp_ctx, out_d, reg_s = res[4][:4]
                        if y_bf is not None:
                            ctxs[-1] = ("chain+rec+proj", *ctxs[-1][1:], head_ctx, samp_ctx)
                        elif REC_TAIL_BWD and ops.gru_seq_bwd_tail_supported(
                                lead[0], rec.hidden_features, self.layers[j + 1].out_features):
                            # the mirror image in the backward: one entry, one launch
                            ctxs[-1] = ("chain+rec+tail", *ctxs[-1][1:], head_ctx, samp_ctx)
                        else:
                            ctxs.append(("chain", j + 1, j + 2, head_ctx, lead))
                            ctxs.append(("layer", j + 2, samp_ctx))
                        final_state.extend([state0[j + 1], ()])
                        reg = add_reg(reg, reg_s)
                        x = out_d
                        i = n
                    continue
                lead = x.shape[:-1]
                x2 = x.reshape(-1, x.shape[-1])
                cctx, y = dense_chain.forward_train(
                    self.layers[i:j], x2 if x2.is_contiguous() else x2.contiguous(),
                    upstream_needs)
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

    def replay_with_bootstrap(self, state0, x_seq, done_seq, extras_seq, last_obs):
        """`replay` plus the value estimate of `last_obs` (the bootstrap V(s_T) of
        ppo.py:433-437) for the network `[Normalizer, PPOAdapter]` whose VALUE port is a
        stateless chain of Dense layers (the action port may be recurrent): the bootstrap
        observation is the value port's step T — normalised into the same buffer as the
        sequence, it rides in the value chain's launch instead of costing a chain launch of its
        own on the critical path (`forward_value` after the replay: at C4 a 1024-row launch
        per gradient step).  The chain's backward then covers the first T steps' rows only.
        Returns (ctx, out, reg, final_state, last_values), or None when the pattern does
        not apply (the caller then takes `replay` + `forward_value`)."""
        import torch

        from . import dense_chain
        from .adapter import PPOAdapter
        from .feedforward import Dense
        from .normalizer import Normalizer
        from ..tree import tree_leaves

        if not BOOTSTRAP_IN_CHAIN or config.compute_dtype() != "bf16" or len(self.layers) != 2:
            return None
        norm, adapter = self.layers
        if not (type(norm) is Normalizer and isinstance(adapter, PPOAdapter)
                and type(adapter.value) is Sequential and adapter.value.layers
                and all(type(l) is Dense for l in adapter.value.layers)):
            return None
        if not (isinstance(x_seq, torch.Tensor) and x_seq.dim() == 3 and x_seq.is_cuda
                and isinstance(last_obs, torch.Tensor) and last_obs.dim() == 2
                and last_obs.shape == x_seq.shape[1:] and isinstance(norm.mean.value, torch.Tensor)
                and x_seq.dtype == torch.float32 and extras_seq is not None):
            return None
        if any(isinstance(t, torch.Tensor) for t in tree_leaves(state0[1]["value"])):
            return None
        if not dense_chain._fusable(adapter.value.layers, x_seq.shape[0] * x_seq.shape[1]):
            return None
        T, B, K = x_seq.shape
        x_ext = torch.empty(T + 1, B, K, dtype=torch.float32, device=x_seq.device)
        cnt = norm.counter.value
        from .. import ops

        ops.normalize_fwd_tail(x_seq if x_seq.is_contiguous() else x_seq.contiguous(),
                               last_obs if last_obs.is_contiguous() else last_obs.contiguous(),
                               norm.mean.value, norm.M2.value, cnt, norm.epsilon, x_ext)
        a_ctx, out, reg, fs = adapter.replay(state0[1], x_ext[:T], done_seq, extras_seq[1],
                                             need_input_grad=False, x_value=x_ext)
        v_ext = out.value_estimates  # [T + 1, B(, n)]: the value port saw the extra step
        out = type(out)(actions=out.actions, loglikelihoods=out.loglikelihoods,
                        value_estimates=v_ext[:T])
        ctxs = [("layer", 0, False), ("layer", 1, a_ctx)]
        return ctxs, out, reg, [(), fs], v_ext[T]

    def replay_backward(self, ctxs, g_out, g_reg):
        g = g_out
        for entry in reversed(ctxs):
            if entry[0] == "chain+rec+proj":
                # sampler, head, BPTT and the projection's backward in one launch; what is left
                # of the chain in front is the dW of its one relu layer
                _, i, j, cctx, lead, rctx, head_ctx, samp_ctx = entry
                rec, front = self.layers[j], self.layers[i]
                dz0_bf = rec.replay_backward_proj_tail(rctx, self.layers[j + 1], head_ctx,
                                                       self.layers[j + 2], samp_ctx, g, g_reg)
                ops.dense_bwd_dw_grouped_bf16(
                    [(cctx[0][0][0], dz0_bf, front.kernel.grad,
                      front.bias.grad if front.bias is not None else None)], accumulate=True)
                return None
            if entry[0] in ("chain+rec", "chain+rec+tail"):
                from . import dense_chain

                _, i, j, cctx, lead, rctx = entry[:6]
                rec = self.layers[j]
                if entry[0] == "chain+rec+tail":
                    dgi = rec.replay_backward_tail(rctx, self.layers[j + 1], entry[6],
                                                   self.layers[j + 2], entry[7], g, g_reg)
                else:
                    dgi = rec.replay_backward(rctx, g, g_reg)
                d2 = dgi.reshape(-1, dgi.shape[-1])
                gi = dense_chain.backward(list(self.layers[i:j]) + [rec.chain_projection()], cctx,
                                          d2 if d2.is_contiguous() else d2.contiguous())
                g = None if gi is None else gi.view(*lead, gi.shape[-1])
            elif entry[0] == "chain":
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


def _resolve_components(kind: str, modules, kwargs) -> dict:
    """containers.py:70-80 / utils.py `_resolve_components`: a positional dict or
    keyword arguments, not both, at least one."""
    if modules is not None and kwargs:
        raise ValueError(f"{kind}: pass either a positional dict or keyword arguments, not both")
    components = modules if modules is not None else kwargs
    if not components:
        raise ValueError(f"{kind} requires at least one component")
    return dict(components)


def _add_tree(a, b):
    from ..tree import tree_map

    if a is None:
        return b
    if b is None:
        return a
    return tree_map(lambda u, v: u + v, a, b)


class _Keyed(StatefulModule):
    """Shared routing of the dict-of-sub-modules containers: carry state, rollout
    extras and metrics are dicts keyed by component name; regularisers are summed
    (containers.py:86-118, 145-176; utils.py Merge / Map)."""

    _KIND = "_Keyed"
    _PER_KEY_INPUT = False  # True: component k sees x[k]; False: every component sees x

    def __init__(self, modules: dict | None = None, /, **kwargs: StatefulModule):
        self.components = _resolve_components(self._KIND, modules, kwargs)

    def _combine(self, outputs: dict):
        raise NotImplementedError

    def _split_grad(self, g_out, outputs_meta: dict) -> dict:
        raise NotImplementedError

    def __call__(self, state, x, rollout_extras=None) -> StatefulModuleOutput:
        new_state, new_extras, outputs, metrics = {}, {}, {}, {}
        reg = zero_scalar(_device_of(x))
        for key, component in self.components.items():
            child_extras = None if rollout_extras is None else rollout_extras[key]
            out = component(state[key], x[key] if self._PER_KEY_INPUT else x, child_extras)
            new_state[key] = out.next_state
            new_extras[key] = out.rollout_extras
            outputs[key] = out.output
            reg = add_reg(reg, out.regularization_loss)
            metrics[key] = out.metrics
        return StatefulModuleOutput(new_state, self._combine(outputs), reg, metrics, new_extras)

    def initialize_state(self, batch_size: int) -> dict:
        return {k: c.initialize_state(batch_size) for k, c in self.components.items()}

    def reset_state(self, prev_state: dict) -> dict:
        return {k: c.reset_state(prev_state[k]) for k, c in self.components.items()}

    def update_statistics(self, rollout_extras: Any) -> None:
        for key, component in self.components.items():
            component.update_statistics(rollout_extras[key])

    # ---- training protocol ---------------------------------------------------------
    def replay(self, state0, x_seq, done_seq, extras_seq, need_input_grad=True):
        ctxs, outputs, final_state = {}, {}, {}
        reg = None
        for key, component in self.components.items():
            child_extras = None if extras_seq is None else extras_seq[key]
            ctx, out, r, fs = component.replay(
                state0[key], x_seq[key] if self._PER_KEY_INPUT else x_seq, done_seq,
                child_extras, need_input_grad=need_input_grad)
            ctxs[key] = ctx
            outputs[key] = out
            final_state[key] = fs
            reg = add_reg(reg, r)
        meta = {k: (v.shape[-1] if hasattr(v, "shape") else None) for k, v in outputs.items()}
        return (ctxs, meta), self._combine(outputs), reg, final_state

    def replay_backward(self, ctx, g_out, g_reg):
        ctxs, meta = ctx
        grads = self._split_grad(g_out, meta)
        g_in = {} if self._PER_KEY_INPUT else None
        for key, component in self.components.items():
            g = component.replay_backward(ctxs[key], grads[key], g_reg)
            if self._PER_KEY_INPUT:
                g_in[key] = g
            else:
                g_in = _add_tree(g_in, g)
        if self._PER_KEY_INPUT and all(v is None for v in g_in.values()):
            return None
        return g_in


class Concat(_Keyed):
    """containers.py:55-122 — per-key dispatch + concatenation along the last axis:
    dict input, single-tensor output."""

    _KIND = "Concat"
    _PER_KEY_INPUT = True

    def _combine(self, outputs: dict):
        import torch

        return torch.cat(list(outputs.values()), dim=-1)

    def _split_grad(self, g_out, meta: dict) -> dict:
        out, o = {}, 0
        for k, n in meta.items():
            out[k] = g_out[..., o:o + n]
            o += n
        return out


class Parallel(_Keyed):
    """containers.py:125-180 — several sub-modules on the SAME input, outputs as a
    dict keyed by sub-module name."""

    _KIND = "Parallel"

    def _combine(self, outputs: dict):
        return outputs

    def _split_grad(self, g_out, meta: dict) -> dict:
        return {k: g_out[k] for k in meta}


class Splitter(StatefulModule):
    """containers.py:183-218 — named slices of the last axis, in keyword order; excess
    input features are ignored (plain slicing semantics)."""

    def __init__(self, **sizes: int):
        if not sizes:
            raise ValueError("Splitter requires at least one named slice")
        for k, v in sizes.items():
            if v <= 0:
                raise ValueError(f"slice size for {k!r} must be positive, got {v}")
        self._sizes = dict(sizes)

    def _split(self, x):
        out, o = {}, 0
        for key, size in self._sizes.items():
            out[key] = x[..., o:o + size]
            o += size
        return out

    def __call__(self, state, x, rollout_extras=None) -> StatefulModuleOutput:
        return StatefulModuleOutput((), self._split(x), zero_scalar(x.device), {}, None)

    def replay(self, state0, x_seq, done_seq, extras_seq, need_input_grad=True):
        return (need_input_grad, x_seq.shape[-1]), self._split(x_seq), None, ()

    def replay_backward(self, ctx, g_out, g_reg):
        import torch

        need, width = ctx
        if not need:
            return None
        parts = []
        ref = next(g for g in g_out.values() if g is not None)
        for key, size in self._sizes.items():
            g = g_out.get(key)
            parts.append(g if g is not None
                         else torch.zeros(*ref.shape[:-1], size, dtype=ref.dtype, device=ref.device))
        used = sum(self._sizes.values())
        if used < width:
            parts.append(torch.zeros(*ref.shape[:-1], width - used, dtype=ref.dtype,
                                     device=ref.device))
        return torch.cat(parts, dim=-1)
