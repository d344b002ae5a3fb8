"""Dense layer (counterpart of `nnx_ppo/networks/feedforward.py:13-51`):
`y = act(x @ W + b)` with `W: [in, out]`, run by the MFMA GEMM kernels of
csrc/dense.hip (forward, dX, dW + db)."""
from __future__ import annotations

from typing import Any, Optional

import torch

from .. import config, ops
from . import activations, initializers
from .types import Parameter, Rngs, StatefulModule, StatefulModuleOutput, zero_scalar


def _rows(x, width: int):
    """`[..., width]` -> contiguous `[M, width]` (a slice handed over by a routing
    container, e.g. `Splitter`, is a strided view)."""
    x2 = x.reshape(-1, width)
    return x2 if x2.is_contiguous() else x2.contiguous()


class Dense(StatefulModule):
    def __init__(self, in_features: int, out_features: int, rngs: Rngs,
                 activation: Any = None, *, kernel_init=None, bias_init=None,
                 use_bias: bool = True):
        self.in_features = in_features
        self.out_features = out_features
        self.activation = activation
        self.act_code = activations.resolve(activation)
        gen = rngs.generator()
        kernel_init = kernel_init or initializers.lecun_normal()
        self.kernel = Parameter(kernel_init(gen, (in_features, out_features)))
        self.bias = None
        if use_bias:
            self.bias = Parameter((bias_init or initializers.zeros)(gen, (out_features,)))

    def _fwd(self, x2: torch.Tensor, want_aux: bool):
        b = self.bias.data if self.bias is not None else None
        if want_aux and self.act_code == ops.ACT_SWISH:
            y, pre = ops.dense_fwd(x2, self.kernel.data, b, self.act_code, want_preact=True)
            return y, pre
        y = ops.dense_fwd(x2, self.kernel.data, b, self.act_code)
        return y, y

    def __call__(self, state, x: torch.Tensor, rollout_extras: Any = None) -> StatefulModuleOutput:
        lead = x.shape[:-1]
        if config.compute_dtype() == "bf16":
            from . import dense_chain

            y = dense_chain.forward_infer([self], _rows(x, self.in_features))
        else:
            y, _ = self._fwd(_rows(x, self.in_features), want_aux=False)
        y = y.view(*lead, self.out_features)
        return StatefulModuleOutput(state, y, zero_scalar(x.device), {}, None)

    # ---- training protocol ----------------------------------------------------
    def replay(self, state0, x_seq, done_seq, extras_seq, need_input_grad=True):
        lead = x_seq.shape[:-1]
        x2 = _rows(x_seq, self.in_features)
        if config.compute_dtype() == "bf16":
            from . import dense_chain

            cctx, y = dense_chain.forward_train([self], x2, need_input_grad)
            return ("bf16", cctx, lead), y.view(*lead, self.out_features), None, state0
        y, aux = self._fwd(x2, want_aux=True)
        ctx = (x2, aux, lead, need_input_grad)
        return ctx, y.view(*lead, self.out_features), None, state0

    def replay_backward(self, ctx, g_out, g_reg):
        if ctx[0] == "bf16":
            from . import dense_chain

            _, cctx, lead = ctx
            g_in = dense_chain.backward([self], cctx, _rows(g_out, self.out_features))
            return None if g_in is None else g_in.view(*lead, self.in_features)
        x2, aux, lead, need_input_grad = ctx
        g2 = g_out.reshape(-1, self.out_features)
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        ops.dense_bwd_dw(x2, g2, aux, self.kernel.grad,
                         self.bias.grad if self.bias is not None else None, self.act_code,
                         accumulate=True)
        if not need_input_grad:
            return None
        g_x = ops.dense_bwd_dx(g2, aux, self.kernel.data, self.act_code)
        return g_x.view(*lead, self.in_features)
