"""Activation markers in the role of `nnx.relu / nnx.swish / nnx.tanh`
(`nnx_ppo/networks/factories.py:107-110`).  Layers resolve them to the kernel's
activation code; calling one applies the function with torch (host-side
convenience for user code, not used by the kernels)."""
from __future__ import annotations

import torch

from .. import ops


class _Activation:
    def __init__(self, name: str, code: int, fn):
        self.name, self.code, self._fn = name, code, fn

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self._fn(x)

    def __repr__(self):
        return f"<activation {self.name}>"


relu = _Activation("relu", ops.ACT_RELU, torch.relu)
tanh = _Activation("tanh", ops.ACT_TANH, torch.tanh)
swish = _Activation("swish", ops.ACT_SWISH, lambda x: x * torch.sigmoid(x))
silu = swish


def resolve(act) -> int:
    """Activation argument (None | str | marker) -> MI_ACT_* code."""
    if act is None:
        return ops.ACT_NONE
    if isinstance(act, _Activation):
        return act.code
    if isinstance(act, str):
        if act in ops.ACT_CODES:
            return ops.ACT_CODES[act]
        raise KeyError(act)
    raise TypeError(
        f"unsupported activation {act!r}: use None, 'relu' | 'tanh' | 'swish', or "
        "nnx_ppo_amd.networks.activations.{relu,tanh,swish}")
