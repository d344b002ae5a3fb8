"""The network plugin protocol (counterpart of `nnx_ppo/networks/types.py`).

`StatefulModule` keeps the reference's single-step interface verbatim
(`__call__(state, obs, rollout_extras=None) -> StatefulModuleOutput`,
`initialize_state`, `reset_state`, `update_statistics`; types.py:39-113) — this
is what the rollout, eval and any user code call.

The reference obtains gradients by tracing that same `__call__` with `nnx.grad`
inside a T-step scan (`ppo.py:301-312,415-431`).  There is no tracing compiler
here, so the training path is an explicit *sequence-level* protocol with a
hand-written backward:

    ctx, out_seq, reg_seq, final_state = m.replay(state0, x_seq, done_seq, extras_seq)
    g_x_seq = m.replay_backward(ctx, g_out_seq, g_reg)

`x_seq` leaves are time-major `[T, B, ...]`.  A module processes the WHOLE
sequence before the next one runs: inside a feed-forward composition each
layer at time t depends only on the previous layer at time t and on its own
carry from t-1, so layer-by-layer evaluation gives exactly the scan's values
while letting stateless layers run one time-batched `[T*B, K]` GEMM and
recurrent layers run one persistent T-loop kernel.  `done_seq[t]` resets a
module's carry after step t (`ppo.py:411-418`).  Parameter gradients are
accumulated into `Parameter.grad` (views of one flat arena).
"""
from __future__ import annotations

import dataclasses
from typing import Any, Iterator, Union

import numpy as np
import torch

from ..tree import TreeDataclass, tree_leaves

ModuleState = Any  # any pytree: (), (h, c), dict, ...


@dataclasses.dataclass(frozen=True)
class PPONetworkOutput(TreeDataclass):
    """types.py:14-26."""

    actions: Any
    loglikelihoods: Any
    value_estimates: Any


@dataclasses.dataclass(frozen=True)
class StatefulModuleOutput(TreeDataclass):
    """types.py:29-36."""

    next_state: ModuleState
    output: Any
    regularization_loss: Any  # scalar or [batch]
    metrics: dict
    rollout_extras: Any = None


class Parameter:
    """Trainable fp32 tensor (the `nnx.Param` role).  After
    `ParamArena.bind()` `.data` / `.grad` are views into flat arenas."""

    def __init__(self, value):
        if isinstance(value, np.ndarray):
            value = torch.from_numpy(np.ascontiguousarray(value, dtype=np.float32))
        self.data: torch.Tensor = value.to(torch.float32).contiguous()
        self.grad: torch.Tensor | None = None

    @property
    def shape(self):
        return self.data.shape

    def numel(self) -> int:
        return self.data.numel()

    def __repr__(self):
        return f"Parameter{tuple(self.data.shape)}"


class Variable:
    """Non-trainable module variable (e.g. normaliser statistics)."""

    def __init__(self, value):
        if isinstance(value, np.ndarray):
            value = torch.from_numpy(np.ascontiguousarray(value))
        self.value: torch.Tensor = value

    def get_value(self) -> torch.Tensor:
        return self.value

    def set_value(self, v: torch.Tensor) -> None:
        self.value = v


class Rngs:
    """Seed bundle in the role of `nnx.Rngs(seed, **streams)`: a numpy Generator
    for parameter init and integer seeds for named noise streams."""

    def __init__(self, default: int = 0, **streams: int):
        self.default_seed = int(default)
        self.streams = {k: int(v) for k, v in streams.items()}
        self._np = np.random.default_rng(self.default_seed)
        self._n_streams_handed_out = 0

    def generator(self) -> np.random.Generator:
        return self._np

    def stream_seed(self, name: str = "action_sampling") -> int:
        """A fresh 64-bit seed for a noise-consuming module (each call differs)."""
        base = self.streams.get(name, self.default_seed)
        self._n_streams_handed_out += 1
        z = (base * 0x9E3779B97F4A7C15 + self._n_streams_handed_out * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z ^= z >> 31
        return z & (2**63 - 1)


class StatefulModule:
    """types.py:39-113 + the sequence-level training protocol (module docstring)."""

    # ---- reference interface -------------------------------------------------
    def __call__(self, module_state: ModuleState, obs: Any,
                 rollout_extras: Any = None) -> StatefulModuleOutput:
        raise NotImplementedError

    def initialize_state(self, batch_size: int) -> ModuleState:
        return ()

    def reset_state(self, prev_state: ModuleState) -> ModuleState:
        return prev_state

    def update_statistics(self, rollout_extras: Any) -> None:
        del rollout_extras
        return None

    def forward_value(self, module_state: ModuleState, obs: Any) -> Any:
        """Value estimates only, for the bootstrap query V(s_T) of the loss
        (`ppo.py:433-437`).  The reference evaluates the whole network there and
        XLA drops the unused action branch; containers override this to skip the
        action port explicitly (no sampler call, so no noise offset is consumed)."""
        return self(module_state, obs).output.value_estimates

    # ---- sequence-level training protocol -------------------------------------
    def replay(self, state0: ModuleState, x_seq: Any, done_seq: torch.Tensor,
               extras_seq: Any, need_input_grad: bool = True):
        """Returns (ctx, out_seq, reg_seq, final_state).  `reg_seq` is None (zero),
        or a `[T, B]` tensor of per-step regularisation losses."""
        raise NotImplementedError(
            f"{type(self).__name__} does not implement the sequence-level training "
            "protocol (replay / replay_backward); see nnx_ppo_amd/networks/types.py")

    def replay_backward(self, ctx: Any, g_out: Any, g_reg: float) -> Any:
        raise NotImplementedError(f"{type(self).__name__}.replay_backward")

    # ---- module tree ------------------------------------------------------------
    def _children(self) -> Iterator[tuple[str, "StatefulModule"]]:
        for name, v in vars(self).items():
            if isinstance(v, StatefulModule):
                yield name, v
            elif isinstance(v, (list, tuple)):
                for i, c in enumerate(v):
                    if isinstance(c, StatefulModule):
                        yield f"{name}.{i}", c
            elif isinstance(v, dict):
                for k, c in v.items():
                    if isinstance(c, StatefulModule):
                        yield f"{name}.{k}", c

    def modules(self) -> list["StatefulModule"]:
        seen: dict[int, StatefulModule] = {}

        def rec(m):
            if id(m) in seen:
                return
            seen[id(m)] = m
            for _, c in m._children():
                rec(c)

        rec(self)
        return list(seen.values())

    def named_parameters(self) -> list[tuple[str, Parameter]]:
        out: list[tuple[str, Parameter]] = []
        seen: set[int] = set()

        def rec(m, prefix):
            for name, v in vars(m).items():
                if isinstance(v, Parameter) and id(v) not in seen:
                    seen.add(id(v))
                    out.append((prefix + name, v))
            for cname, c in m._children():
                rec(c, f"{prefix}{cname}.")

        rec(self, "")
        return out

    def parameters(self) -> list[Parameter]:
        return [p for _, p in self.named_parameters()]

    def variables(self) -> list[Variable]:
        out = []
        for m in self.modules():
            for v in vars(m).values():
                if isinstance(v, Variable):
                    out.append(v)
                elif isinstance(v, dict):
                    out.extend(x for x in v.values() if isinstance(x, Variable))
        return out

    def to(self, device) -> "StatefulModule":
        device = torch.device(device)
        for p in self.parameters():
            p.data = p.data.to(device)
            if p.grad is not None:
                p.grad = p.grad.to(device)
        for m in self.modules():
            m._to_device(device)
            m._device = device
        return self

    def _to_device(self, device) -> None:
        for name, v in list(vars(self).items()):
            if isinstance(v, Variable):
                v.value = _tree_to(v.value, device)
            elif isinstance(v, torch.Tensor):
                setattr(self, name, v.to(device))

    @property
    def device(self) -> torch.device:
        d = getattr(self, "_device", None)
        if d is not None:
            return d
        ps = self.parameters()
        if ps:
            return ps[0].data.device
        for v in self.variables():
            leaves = tree_leaves(v.value)
            if leaves:
                return leaves[0].device
        return torch.device("cpu")

    # `.eval()` / `.train()` flip `deterministic` on samplers, as flax's
    # Module.eval()/train() do (factories_test.py:64-74).
    def eval(self) -> "StatefulModule":
        for m in self.modules():
            if hasattr(m, "deterministic"):
                m.deterministic = True
        return self

    def train(self) -> "StatefulModule":
        for m in self.modules():
            if hasattr(m, "deterministic"):
                m.deterministic = False
        return self


def _tree_to(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device)
    if isinstance(x, dict):
        return {k: _tree_to(v, device) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_tree_to(v, device) for v in x)
    return x


MetricsKey = Union[str, int]


_ZERO_SCALARS: dict[str, torch.Tensor] = {}


def zero_scalar(device) -> torch.Tensor:
    """Cached fp32 scalar 0 per device — the `jp.array(0.0)` regularisation loss
    of layers that have none (`feedforward.py:51`).  Containers recognise it by
    identity and skip the add, so summing regularisers costs no launches."""
    key = str(device)
    z = _ZERO_SCALARS.get(key)
    if z is None:
        z = torch.zeros((), dtype=torch.float32, device=device)
        _ZERO_SCALARS[key] = z
    return z


def add_reg(a, b):
    """a + b where either may be None, the cached zero scalar, or a cached all-zero
    constant (`envs.constants.constant`, e.g. the `[B]` zeros a recurrent layer returns,
    `recurrent.py:114`): those cost no launch — the other operand is returned when the sum
    would have its shape."""
    from ..envs.constants import is_zero_constant

    if b is None or any(b is z for z in _ZERO_SCALARS.values()):
        return a
    if a is None or any(a is z for z in _ZERO_SCALARS.values()):
        return b
    if is_zero_constant(b) and (a.shape == b.shape or b.dim() == 0):
        return a
    if is_zero_constant(a) and (a.shape == b.shape or a.dim() == 0):
        return b
    return a + b


# Epoch of the trainable parameters: bumped whenever they may have changed
# (optimiser update, start of an iteration).  bf16 weight shadows compare it to
# decide when to refresh.
_PARAM_EPOCH = [0]


def param_epoch() -> int:
    return _PARAM_EPOCH[0]


def bump_param_epoch() -> None:
    _PARAM_EPOCH[0] += 1
