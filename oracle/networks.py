"""Network modules of the oracle (torch-CPU, fp64 by default, autograd for
gradients).  TEST INFRASTRUCTURE — see oracle/__init__.py.

Single-step `StatefulModule` semantics restated from the reference:
  types        nnx_ppo/networks/types.py:29-113
  Dense        nnx_ppo/networks/feedforward.py:42-51
  Sequential   nnx_ppo/networks/containers.py:18-52
  PPOAdapter   nnx_ppo/networks/adapter.py:75-133
  Normalizer   nnx_ppo/networks/normalizer.py:63-136
  Sampler      nnx_ppo/networks/sampling_layers.py:82-147
  Flattener    nnx_ppo/networks/utils.py:65-116
  LSTM         nnx_ppo/networks/recurrent.py:89-161 (cell: flax.nnx.LSTMCell, unpinned)
  GRU          no reference code; contract of the LSTM wrapper, cell arithmetic of
               flax.nnx.GRUCell (unpinned)
Unlike the product there is no sequence-level protocol here: the oracle scans
time step by step exactly as the reference does and lets autograd differentiate.
"""
from __future__ import annotations

import math
from typing import Any

import numpy as np
import torch

from . import philox

DTYPE = torch.float64


class Out:
    def __init__(self, next_state, output, regularization_loss, metrics, rollout_extras=None):
        self.next_state = next_state
        self.output = output
        self.regularization_loss = regularization_loss
        self.metrics = metrics
        self.rollout_extras = rollout_extras


class PPOOut:
    def __init__(self, actions, loglikelihoods, value_estimates):
        self.actions = actions
        self.loglikelihoods = loglikelihoods
        self.value_estimates = value_estimates


def _t(x, dtype=DTYPE):
    if isinstance(x, torch.Tensor):
        return x.detach().to("cpu", dtype)
    return torch.as_tensor(np.asarray(x), dtype=dtype)


class Module:
    def __call__(self, state, obs, extras=None) -> Out:
        raise NotImplementedError

    def forward_value(self, state, obs):
        """Value estimates of the bootstrap query (ppo.py:433-437).  Only the value
        branch is evaluated — the reference's action sample at this point is dead
        code — so no sampler noise offset is consumed (scheme shared with the
        product; RNG stream alignment is unpinned by the reference)."""
        return self(state, obs).output.value_estimates

    def initialize_state(self, batch_size: int):
        return ()

    def reset_state(self, prev):
        return prev

    def update_statistics(self, extras) -> None:
        return None

    def children(self):
        return []

    def own_parameters(self):
        return []

    def parameters(self):
        ps = list(self.own_parameters())
        for c in self.children():
            ps.extend(c.parameters())
        return ps

    def modules(self):
        out = [self]
        for c in self.children():
            out.extend(c.modules())
        return out

    def eval(self):
        for m in self.modules():
            if hasattr(m, "deterministic"):
                m.deterministic = True
        return self

    def train(self):
        for m in self.modules():
            if hasattr(m, "deterministic"):
                m.deterministic = False
        return self


def _act(name):
    # factories.py:107-110: relu | swish | tanh
    return {None: None, "none": None, "relu": torch.relu, "tanh": torch.tanh,
            "swish": lambda z: z * torch.sigmoid(z)}[name]


class Dense(Module):
    """feedforward.py:42-51: y = act(x @ W + b), W: [in, out]."""

    def __init__(self, kernel, bias, activation=None, dtype=DTYPE):
        self.kernel = _t(kernel, dtype).requires_grad_(True)
        self.bias = None if bias is None else _t(bias, dtype).requires_grad_(True)
        self.activation = activation

    def own_parameters(self):
        return [self.kernel] + ([self.bias] if self.bias is not None else [])

    def __call__(self, state, x, extras=None):
        y = x.to(self.kernel.dtype) @ self.kernel
        if self.bias is not None:
            y = y + self.bias
        f = _act(self.activation)
        if f is not None:
            y = f(y)
        return Out(state, y, torch.zeros((), dtype=y.dtype), {}, None)


class Sequential(Module):
    """containers.py:18-52."""

    def __init__(self, layers):
        self.layers = list(layers)

    def children(self):
        return self.layers

    def __call__(self, state, obs, extras=None):
        new_state, new_extras, metrics = [], [], {}
        x = obs
        reg = torch.zeros((), dtype=DTYPE)
        for i, (layer, st) in enumerate(zip(self.layers, state)):
            out = layer(st, x, None if extras is None else extras[i])
            new_state.append(out.next_state)
            new_extras.append(out.rollout_extras)
            x = out.output
            reg = reg + out.regularization_loss
            metrics[len(metrics)] = out.metrics
        return Out(new_state, x, reg, metrics, new_extras)

    def forward_value(self, state, obs):
        x = obs
        for layer, st in zip(self.layers[:-1], state[:-1]):
            x = layer(st, x, None).output
        return self.layers[-1].forward_value(state[-1], x)

    def initialize_state(self, batch_size):
        return [l.initialize_state(batch_size) for l in self.layers]

    def reset_state(self, prev):
        return [l.reset_state(s) for l, s in zip(self.layers, prev)]

    def update_statistics(self, extras):
        for l, e in zip(self.layers, extras):
            l.update_statistics(e)


class PPOAdapter(Module):
    """adapter.py:75-133."""

    def __init__(self, action, value):
        self.action, self.value = action, value

    def children(self):
        return [self.action, self.value]

    @staticmethod
    def _pick(tree, key):
        """adapter.py:92-98: map over a tree whose leaves are sampler dicts."""
        if isinstance(tree, dict) and {"action", "log_likelihood"} <= set(tree):
            return tree[key]
        if isinstance(tree, dict):
            return {k: PPOAdapter._pick(v, key) for k, v in tree.items()}
        if isinstance(tree, (list, tuple)):
            return type(tree)(PPOAdapter._pick(v, key) for v in tree)
        raise TypeError("the action port must emit sampler dicts")

    @staticmethod
    def _squeeze(val):
        return val.squeeze(-1) if val.shape and val.shape[-1] == 1 else val  # adapter.py:55-58

    def __call__(self, state, x, extras=None):
        a_re = None if extras is None else extras["action"]
        v_re = None if extras is None else extras["value"]
        a = self.action(state["action"], x, a_re)
        v = self.value(state["value"], x, v_re)
        return Out({"action": a.next_state, "value": v.next_state},
                   PPOOut(self._pick(a.output, "action"), self._pick(a.output, "log_likelihood"),
                          _map(self._squeeze, v.output)),
                   a.regularization_loss + v.regularization_loss,
                   {"action": a.metrics, "value": v.metrics},
                   {"action": a.rollout_extras, "value": v.rollout_extras})

    def forward_value(self, state, x):
        return _map(self._squeeze, self.value(state["value"], x, None).output)

    def initialize_state(self, batch_size):
        return {"action": self.action.initialize_state(batch_size),
                "value": self.value.initialize_state(batch_size)}

    def reset_state(self, prev):
        return {"action": self.action.reset_state(prev["action"]),
                "value": self.value.reset_state(prev["value"])}

    def update_statistics(self, extras):
        self.action.update_statistics(extras["action"])
        self.value.update_statistics(extras["value"])


def _leaves(tree):
    if tree is None:
        return []
    if isinstance(tree, dict):
        out = []
        for k in sorted(tree):
            out.extend(_leaves(tree[k]))
        return out
    if isinstance(tree, (list, tuple)):
        out = []
        for v in tree:
            out.extend(_leaves(v))
        return out
    return [tree]


def _map(fn, tree, *rest):
    if tree is None:
        return None
    if isinstance(tree, dict):
        return {k: _map(fn, tree[k], *[r[k] for r in rest]) for k in tree}
    if isinstance(tree, (list, tuple)):
        return type(tree)(_map(fn, v, *[r[i] for r in rest]) for i, v in enumerate(tree))
    return fn(tree, *rest)


class Normalizer(Module):
    """normalizer.py:35-136.  `shape`: int | tuple | dict of those."""

    def __init__(self, shape, dtype=DTYPE):
        mk = lambda s: torch.zeros((s,) if isinstance(s, int) else tuple(s), dtype=dtype)
        isshape = lambda s: isinstance(s, int) or (
            isinstance(s, (tuple, list)) and all(isinstance(v, int) for v in s))
        if isshape(shape):
            self.mean, self.M2 = mk(shape), mk(shape)
        else:
            self.mean = {k: mk(v) for k, v in shape.items()}
            self.M2 = {k: mk(v) for k, v in shape.items()}
        self.counter = torch.zeros((), dtype=dtype)
        self.epsilon = 1e-6
        self.dtype = dtype

    def _std(self):
        if float(self.counter) > 0:  # normalizer.py:76-81,92-96
            return _map(lambda m2: torch.sqrt(torch.clamp(m2 / self.counter, min=self.epsilon)),
                        self.M2)
        return _map(lambda m2: torch.full_like(m2, 10.0), self.M2)

    def __call__(self, state, x, extras=None):
        x = _map(lambda v: v.to(self.dtype), x)
        out = _map(lambda v, m, s: (v - m) / s, x, self.mean, self._std())
        return Out((), out, torch.zeros((), dtype=self.dtype), {}, x)

    def update_statistics(self, extras):
        """normalizer.py:98-136 (two-pass batch moments, then the pairwise merge)."""
        leaves = _leaves(extras)
        n = leaves[0].shape[0] * leaves[0].shape[1]
        new_count = self.counter + n
        frac = n / new_count

        def merge(v, mean, m2):
            flat = v.to(self.dtype).reshape((-1,) + tuple(v.shape[2:]))
            bm = flat.mean(dim=0)
            bm2 = torch.square(flat - bm).sum(dim=0)
            delta = bm - mean
            return (mean + delta * frac,
                    m2 + bm2 + (delta * delta) * self.counter * n / new_count)

        if isinstance(self.mean, dict):
            for k in self.mean:
                self.mean[k], self.M2[k] = merge(extras[k], self.mean[k], self.M2[k])
        else:
            self.mean, self.M2 = merge(extras, self.mean, self.M2)
        self.counter = new_count


class NormalTanhSampler(Module):
    """sampling_layers.py:66-147.  Noise: Philox scheme of oracle/philox.py, one
    `offset` per batch forward.  During a loss replay the product evaluates all T
    steps in one time-batched call, so `begin_replay(T)` makes the next T
    step-wise calls draw their noise as slices of ONE `[T*B, A]` block (same
    element indexing as the device kernel)."""

    def __init__(self, seed: int, entropy_weight: float, min_std: float = 1e-3,
                 std_scale: float = 1.0, dtype=DTYPE):
        self.seed, self.offset = int(seed), 0
        self.entropy_weight, self.min_std, self.std_scale = entropy_weight, min_std, std_scale
        self.deterministic = False
        self.dtype = dtype
        self._block = None
        self.noise_override = None  # callable(B, A) -> (eps, eps2)

    def begin_replay(self, T: int):
        self._block = {"T": T, "t": 0, "noise": None}

    def _draw(self, B, A):
        if self.noise_override is not None:
            e, e2 = self.noise_override(B, A)
            return _t(e, self.dtype), _t(e2, self.dtype)
        e, e2 = philox.normal_pair(self.seed, self.offset, B * A)
        self.offset += 1
        return (torch.from_numpy(e).to(self.dtype).reshape(B, A),
                torch.from_numpy(e2).to(self.dtype).reshape(B, A))

    def _noise(self, B, A):
        if self._block is not None:
            blk = self._block
            if blk["noise"] is None:
                e, e2 = self._draw(blk["T"] * B, A)
                blk["noise"] = (e.reshape(blk["T"], B, A), e2.reshape(blk["T"], B, A))
            t = blk["t"]
            blk["t"] += 1
            e, e2 = blk["noise"][0][t], blk["noise"][1][t]
            if blk["t"] == blk["T"]:
                self._block = None
            return e, e2
        return self._draw(B, A)

    @staticmethod
    def _log_det_jac(z):
        return 2.0 * (math.log(2.0) - z - torch.nn.functional.softplus(-2.0 * z))

    def __call__(self, state, mean_and_std, extras=None):
        A = mean_and_std.shape[-1] // 2
        mean, s = mean_and_std[..., :A], mean_and_std[..., A:]
        std = (torch.nn.functional.softplus(s) + self.min_std) * self.std_scale
        eps, eps2 = self._noise(mean.shape[0], A)
        sampled = mean if self.deterministic else mean + std * eps
        raw = sampled.detach() if extras is None else extras
        action = torch.tanh(raw)
        # _loglikelihood 118-135
        lp = -0.5 * torch.square((raw - mean) / std) - (0.5 * math.log(2.0 * math.pi) + torch.log(std))
        lp = lp - self._log_det_jac(raw)
        ll = lp.sum(-1)
        # _entropy 137-147
        z = mean + std * eps2.detach()
        ent = (0.5 + 0.5 * math.log(2.0 * math.pi) + torch.log(std) + self._log_det_jac(z)).sum(-1)
        return Out((), {"action": action, "log_likelihood": ll}, -self.entropy_weight * ent,
                   {"mu": mean, "sigma": std}, raw)


class Flattener(Module):
    """utils.py:100-108 with preserve_levels=0: leaves in sorted-key order."""

    def __call__(self, state, x, extras=None):
        leaves = _leaves(x)
        out = torch.cat([a.reshape(a.shape[0], -1) for a in leaves], dim=-1)
        return Out((), out, torch.zeros((), dtype=out.dtype), {}, None)


class GRU(Module):
    """GRU StatefulModule obeying the LSTM wrapper's contract (recurrent.py:89-161:
    zeros init, zeros-like reset, reg = zeros(B), extras None, output = new h).
    Cell arithmetic of flax.nnx.GRUCell (PARITY UNPINNED):
        r = sigmoid(x W_ir + b_ir + h W_hr)
        z = sigmoid(x W_iz + b_iz + h W_hz)
        n = tanh(x W_in + b_in + r * (h W_hn + b_hn))
        h' = (1 - z) * n + z * h
    Weights packed as w_i [in, 3H], b_i [3H], w_h [H, 3H], b_hn [H], gate order (r, z, n)."""

    def __init__(self, w_i, b_i, w_h, b_hn, dtype=DTYPE):
        self.w_i = _t(w_i, dtype).requires_grad_(True)
        self.b_i = _t(b_i, dtype).requires_grad_(True)
        self.w_h = _t(w_h, dtype).requires_grad_(True)
        self.b_hn = _t(b_hn, dtype).requires_grad_(True)
        self.H = self.w_h.shape[0]
        self.dtype = dtype

    def own_parameters(self):
        return [self.w_i, self.b_i, self.w_h, self.b_hn]

    def __call__(self, state, x, extras=None):
        h = state
        H = self.H
        gi = x @ self.w_i + self.b_i
        gh = h @ self.w_h
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * (gh[:, 2 * H:] + self.b_hn))
        h2 = (1.0 - z) * n + z * h
        return Out(h2, h2, torch.zeros(x.shape[0], dtype=self.dtype), {}, None)

    def initialize_state(self, batch_size):
        return torch.zeros(batch_size, self.H, dtype=self.dtype)

    def reset_state(self, prev):
        return torch.zeros_like(prev)


class _Keyed(Module):
    """Dict-of-sub-modules routing shared by Concat / Parallel / Merge / Map
    (containers.py:55-180, utils.py:186-326)."""

    per_key_input = False

    def __init__(self, components: dict):
        self.components = dict(components)

    def children(self):
        return list(self.components.values())

    def combine(self, outputs: dict):
        return outputs

    def __call__(self, state, x, extras=None):
        new_state, new_extras, outputs, metrics = {}, {}, {}, {}
        reg = torch.zeros((), dtype=DTYPE)
        for k, c in self.components.items():
            out = c(state[k], x[k] if self.per_key_input else x,
                    None if extras is None else extras[k])
            new_state[k], new_extras[k], outputs[k], metrics[k] = (
                out.next_state, out.rollout_extras, out.output, out.metrics)
            reg = reg + out.regularization_loss
        return Out(new_state, self.combine(outputs), reg, metrics, new_extras)

    def initialize_state(self, batch_size):
        return {k: c.initialize_state(batch_size) for k, c in self.components.items()}

    def reset_state(self, prev):
        return {k: c.reset_state(prev[k]) for k, c in self.components.items()}

    def update_statistics(self, extras):
        for k, c in self.components.items():
            c.update_statistics(extras[k])


class Concat(_Keyed):
    per_key_input = True

    def combine(self, outputs):
        return torch.cat(list(outputs.values()), dim=-1)


class Parallel(_Keyed):
    pass


class Map(_Keyed):
    per_key_input = True


class Merge(_Keyed):
    def combine(self, outputs):
        merged = {}
        for out in outputs.values():
            for k, v in out.items():
                assert k not in merged
                merged[k] = v
        return merged


class Splitter(Module):
    """containers.py:183-218."""

    def __init__(self, sizes: dict):
        self.sizes = dict(sizes)

    def __call__(self, state, x, extras=None):
        out, o = {}, 0
        for k, n in self.sizes.items():
            out[k] = x[..., o:o + n]
            o += n
        return Out((), out, torch.zeros((), dtype=DTYPE), {}, None)


class Filter(Module):
    """utils.py:119-165."""

    def __init__(self, spec: dict):
        self.spec = dict(spec)

    def __call__(self, state, x, extras=None):
        out = {}
        for k, sub in self.spec.items():
            if isinstance(sub, str):
                out[k] = x[sub]
            elif isinstance(sub, tuple):
                v = x
                for p in sub:
                    v = v[p]
                out[k] = v
            else:
                out[k] = sub(x)
        return Out((), out, torch.zeros((), dtype=DTYPE), {}, None)


class Scale(Module):
    """utils.py:168-183."""

    def __init__(self, factor: float):
        self.factor = float(factor)

    def __call__(self, state, x, extras=None):
        return Out(state, _map(lambda v: v * self.factor, x), torch.zeros((), dtype=DTYPE), {},
                   None)


_LSTM_FNS = {0: lambda x: x, 1: torch.relu, 2: torch.tanh, 4: torch.sigmoid}


class LSTM(Module):
    """`nnx_ppo/networks/recurrent.py:16-161`: carry (h, c); zeros — or, with
    `trainable_initial_state`, the learnable `initial_h` / `initial_c` broadcast over the
    batch (85-88, 132-161) — at init and reset; reg = zeros(B), extras None, output = new h.
    Cell arithmetic of flax.nnx.LSTMCell (third-party, PARITY UNPINNED), gate order
    (i, f, g, o), `gate_fn` on i, f, o and `activation_fn` on g and the new cell state:
        a = x W_i + h W_h + b_h ;  c' = gate(a_f) c + gate(a_i) act(a_g) ;  h' = gate(a_o) act(c')"""

    def __init__(self, w_i, w_h, b_h, dtype=DTYPE, initial_h=None, initial_c=None,
                 gate_act: int = 4, cell_act: int = 2):
        self.w_i = _t(w_i, dtype).requires_grad_(True)
        self.w_h = _t(w_h, dtype).requires_grad_(True)
        self.b_h = _t(b_h, dtype).requires_grad_(True)
        self.H = self.w_h.shape[0]
        self.dtype = dtype
        self.initial_h = None if initial_h is None else _t(initial_h, dtype).requires_grad_(True)
        self.initial_c = None if initial_c is None else _t(initial_c, dtype).requires_grad_(True)
        self.gate, self.act = _LSTM_FNS[gate_act], _LSTM_FNS[cell_act]

    def own_parameters(self):
        ps = [self.w_i, self.w_h, self.b_h]
        if self.initial_h is not None:
            ps += [self.initial_h, self.initial_c]
        return ps

    def __call__(self, state, x, extras=None):
        h, c = state
        H = self.H
        a = x @ self.w_i + h @ self.w_h + self.b_h
        i = self.gate(a[:, :H])
        f = self.gate(a[:, H:2 * H])
        g = self.act(a[:, 2 * H:3 * H])
        o = self.gate(a[:, 3 * H:])
        c2 = f * c + i * g
        h2 = o * self.act(c2)
        return Out((h2, c2), h2, torch.zeros(x.shape[0], dtype=self.dtype), {}, None)

    def initialize_state(self, batch_size):
        if self.initial_h is not None:
            return (self.initial_h.expand(batch_size, self.H),
                    self.initial_c.expand(batch_size, self.H))
        z = lambda: torch.zeros(batch_size, self.H, dtype=self.dtype)
        return (z(), z())

    def reset_state(self, prev):
        if self.initial_h is not None:
            return (self.initial_h.expand(prev[0].shape), self.initial_c.expand(prev[1].shape))
        return (torch.zeros_like(prev[0]), torch.zeros_like(prev[1]))


def from_product(net: Any, dtype=DTYPE) -> Module:
    """Build the oracle twin of a product network by duck-typing on class names
    and copying its weights / statistics / noise seeds (no product import)."""
    name = type(net).__name__
    # (MLPActorCritic / GRUActorCritic are Sequentials with a fused evaluation: policy.py)
    if name in ("Sequential", "MLPActorCritic", "GRUActorCritic"):
        return Sequential([from_product(l, dtype) for l in net.layers])
    if name == "PPOAdapter":
        return PPOAdapter(from_product(net.action, dtype), from_product(net.value, dtype))
    if name == "Dense":
        act = {0: None, 1: "relu", 2: "tanh", 3: "swish"}[net.act_code]
        return Dense(net.kernel.data, None if net.bias is None else net.bias.data, act, dtype)
    if name == "Normalizer":
        mean = net.mean.value
        shape = ({k: tuple(v.shape) for k, v in mean.items()} if isinstance(mean, dict)
                 else tuple(mean.shape))
        o = Normalizer(shape, dtype)
        o.mean = _map(lambda v: _t(v, dtype), mean)
        o.M2 = _map(lambda v: _t(v, dtype), net.M2.value)
        o.counter = _t(net.counter.value, dtype).reshape(())
        return o
    if name == "NormalTanhSampler":
        o = NormalTanhSampler(net.seed, net.entropy_weight, net.min_std, net.std_scale, dtype)
        o.deterministic = net.deterministic
        if net.rng_state is not None:
            o.offset = int(net.rng_state[1].item()) + net._pending
        return o
    if name in ("Concat", "Parallel", "Map", "Merge"):
        cls = {"Concat": Concat, "Parallel": Parallel, "Map": Map, "Merge": Merge}[name]
        return cls({k: from_product(c, dtype) for k, c in net.components.items()})
    if name == "Splitter":
        return Splitter(net._sizes)
    if name == "Filter":
        return Filter(net._spec)
    if name == "Scale":
        return Scale(net.factor)
    if name == "Flattener":
        assert net.preserve_levels == 0
        return Flattener()
    if name == "GRU":
        return GRU(net.w_i.data, net.b_i.data, net.w_h.data, net.b_hn.data, dtype)
    if name == "LSTM":
        ih = net.initial_h.data if getattr(net, "trainable_initial_state", False) else None
        ic = net.initial_c.data if getattr(net, "trainable_initial_state", False) else None
        return LSTM(net.w_i.data, net.w_h.data, net.b_h.data, dtype, ih, ic,
                    getattr(net, "gate_act", 4), getattr(net, "cell_act", 2))
    raise NotImplementedError(f"oracle twin of {name}")
