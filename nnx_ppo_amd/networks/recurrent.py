"""Recurrent modules (counterpart of `nnx_ppo/networks/recurrent.py`): `LSTM`, the
reference's recurrent layer (recurrent.py:16-161), and `GRU`.

The reference ships the LSTM wrapper only; BASELINE.json asks for a GRU carry, so
`GRU` obeys that wrapper's StatefulModule contract (recurrent.py:89-161): state
`[B, H]` zeros at `initialize_state`, zeros-like at `reset_state`, output = new
hidden state, `regularization_loss = zeros(B)`, no rollout extras.  The cell is
flax's GRUCell (third-party; arithmetic PARITY UNPINNED):

    r = sigmoid(x W_ir + b_ir + h W_hr)
    z = sigmoid(x W_iz + b_iz + h W_hz)
    n = tanh(x W_in + b_in + r * (h W_hn + b_hn))
    h' = (1 - z) n + z h

Weights are packed `w_i [in, 3H]`, `b_i [3H]`, `w_h [H, 3H]`, `b_hn [H]` with gate
order (r, z, n).  The input projection is one time-batched GEMM; the recurrence
runs in a persistent kernel that keeps the carry in LDS across the T loop
(csrc/gru.hip); BPTT is its mirror plus time-batched GEMMs for the weights."""
from __future__ import annotations

from typing import Any

import torch

from .. import ops
from ..envs.constants import constant
from . import initializers
from .types import Parameter, Rngs, StatefulModule, StatefulModuleOutput


class _Projection:
    """A weight / bias pair of a recurrent layer presented to `dense_chain` as a
    linear Dense layer (bf16 shadows, whole-trunk kernels, grouped dW)."""

    act_code = ops.ACT_NONE

    def __init__(self, kernel: Parameter, bias, in_features: int, out_features: int):
        self.kernel, self.bias = kernel, bias
        self.in_features, self.out_features = in_features, out_features


class GRU(StatefulModule):
    def __init__(self, in_features: int, hidden_features: int, rngs: Rngs, *, kernel_init=None,
                 recurrent_kernel_init=None):
        if hidden_features > 256:
            raise ValueError("GRU: hidden_features <= 256 in this build")
        self.in_features = in_features
        self.hidden_features = hidden_features
        gen = rngs.generator()
        ki = kernel_init or initializers.lecun_normal()
        kr = recurrent_kernel_init or initializers.lecun_normal()
        H = hidden_features
        import numpy as np

        self.w_i = Parameter(np.concatenate([ki(gen, (in_features, H)) for _ in range(3)], axis=1))
        self.b_i = Parameter(np.zeros(3 * H, dtype=np.float32))
        self.w_h = Parameter(np.concatenate([kr(gen, (H, H)) for _ in range(3)], axis=1))
        self.b_hn = Parameter(np.zeros(H, dtype=np.float32))

    def _gi(self, x2: torch.Tensor) -> torch.Tensor:
        if self._mfma():  # bf16 compute: the input projection is a one-layer bf16 trunk
            from . import dense_chain

            return dense_chain.forward_infer([self._proj()], x2)
        return ops.dense_fwd(x2, self.w_i.data, self.b_i.data, ops.ACT_NONE)

    def _proj(self) -> _Projection:
        p = self.__dict__.get("_proj_i")
        if p is None or p.kernel is not self.w_i:
            p = _Projection(self.w_i, self.b_i, self.in_features, 3 * self.hidden_features)
            self.__dict__["_proj_i"] = p
        return p

    def _mfma(self) -> bool:
        """bf16 compute: the recurrent product runs on the matrix cores (gru_mfma.hip)."""
        from .. import config

        return config.compute_dtype() == "bf16" and ops.gru_mfma_ok(self.hidden_features)

    def __call__(self, state: torch.Tensor, x: torch.Tensor,
                 rollout_extras: Any = None) -> StatefulModuleOutput:
        B = x.shape[0]
        gi = self._gi(x.reshape(B, self.in_features)).view(1, B, 3 * self.hidden_features)
        h_out, _, _, _ = ops.gru_seq_fwd(gi, self.w_h.data, self.b_hn.data, state.contiguous(),
                                         None, train=False, mfma=self._mfma())
        h = h_out[0]
        return StatefulModuleOutput(next_state=h, output=h,
                                    regularization_loss=constant((B,), torch.float32, 0.0, x.device),
                                    metrics={}, rollout_extras=None)

    def initialize_state(self, batch_size: int) -> torch.Tensor:
        return torch.zeros(batch_size, self.hidden_features, dtype=torch.float32,
                           device=self.device)

    def reset_state(self, prev_state: torch.Tensor) -> torch.Tensor:
        # read-only cached zeros: the rollout only selects from a reset state
        return constant(prev_state.shape, prev_state.dtype, 0.0, prev_state.device)

    # ---- training protocol --------------------------------------------------------
    def chain_projection(self):
        """The input projection as a Dense-like layer a preceding run of Dense layers can
        take into its own launch (`containers.Sequential.replay`), or None when the
        projection does not run on the bf16 trunk kernels."""
        return self._proj() if self._mfma() else None

    def replay_tail_supported(self, T: int, head, sampler) -> bool:
        """True when `replay(..., tail=(head, sampler, raw_seq))` can run the linear `head`
        Dense and the tanh-Gaussian `sampler` behind this layer inside the sequence launch
        (`mi_gru_seq_fwd_tail_bf16`)."""
        from .feedforward import Dense
        from .sampling_layers import NormalTanhSampler

        return (self._mfma() and type(head) is Dense and type(sampler) is NormalTanhSampler
                and head.act_code == ops.ACT_NONE and head.in_features == self.hidden_features
                and head.bias is not None and sampler.noise_override is None
                and not sampler.deterministic
                and ops.gru_seq_fwd_tail_supported(T, self.hidden_features, head.out_features))

    def replay_proj_supported(self, T: int, B: int, front, head) -> bool:
        """True when `replay(..., tail=..., proj=...)` can evaluate the input projection inside
        the sequence launches too (`mi_gru_seq_fwd_proj_tail_bf16`): `front`, the layer whose
        output this GRU reads, is a relu Dense of this GRU's width."""
        from .feedforward import Dense

        return (type(front) is Dense and front.act_code == ops.ACT_RELU
                and front.out_features == self.in_features == self.hidden_features
                and ops.gru_seq_proj_supported(T, B, self.hidden_features, self.in_features,
                                               head.out_features))

    def replay_front_supported(self, T: int, B: int, front, head) -> bool:
        """... and `front` itself (`replay(proj=(None, (T, B), front, x2))`,
        `mi_gru_seq_fwd_front_proj_tail_bf16`): at most 8 inputs, with a bias, and room in LDS
        for its T x 4 input and output rows."""
        return (front.bias is not None
                and ops.gru_seq_front_supported(T, B, self.hidden_features, front.in_features,
                                                head.out_features))

    def replay(self, state0, x_seq, done_seq, extras_seq, need_input_grad=True, gi_seq=None,
               tail=None, proj=None):
        """`gi_seq` [T, B, 3H]: the input projection, already evaluated by the caller as the
        last layer of the preceding Dense chain (`x_seq` is then unused and
        `replay_backward` returns the gradient w.r.t. `gi_seq`).
        `tail` = (head Dense, sampler, raw actions [T, B, A]) with `gi_seq`
        (`replay_tail_supported`): head and sampler replay ride in the sequence launch; the
        return value gains a fifth element (head ctx as `dense_chain.forward_train` builds it,
        sampler ctx as `NormalTanhSampler.replay` builds it, sampler output dict, reg [T, B])."""
        H = self.hidden_features
        mfma = self._mfma()
        pctx = None
        if proj is not None:
            # `proj` = (y_bf [T*B, H], (T, B)): the bf16 image of this layer's input; gi is
            # evaluated inside the sequence launch (with `tail`)
            from . import dense_chain

            y_bf, (T, B) = proj[:2]
            head, sampler, raw_seq = tail
            pl = self._proj()
            dense_chain.refresh([pl, head])
            A2 = head.out_features
            ex2 = raw_seq.reshape(T * B, A2 // 2)
            if not ex2.is_contiguous():
                ex2 = ex2.contiguous()
            off = sampler._next_offset()
            front_ctx = None
            if y_bf is None:
                # proj = (None, (T, B), front Dense, x2 [T*B, K0]): the relu layer in front is
                # evaluated inside the launch as well; its bf16 images come back
                front, x2f = proj[2], proj[3]
                dense_chain.refresh([front])
                (h_out, h_prev, gates, h_final, ms2, h_bf, ll, reg, x_bf,
                 y_bf) = ops.gru_seq_fwd_front_proj_tail(
                    x2f, front._ff, front.bias.data, pl._ff, self.b_i.data, self.w_h.data,
                    self.b_hn.data, state0.contiguous(), done_seq.contiguous(), head._ff,
                    head.bias.data, A2, ex2, sampler._state(x2f.device), off, T, **sampler._kw())
                front_ctx = ([(x_bf, y_bf, dense_chain._shadows(front)[0])], T * B, False)
            else:
                h_out, h_prev, gates, h_final, ms2, h_bf, ll, reg = ops.gru_seq_fwd_proj_tail(
                    y_bf, pl._ff, self.b_i.data, self.w_h.data, self.b_hn.data,
                    state0.contiguous(), done_seq.contiguous(), head._ff, head.bias.data, A2, ex2,
                    sampler._state(y_bf.device), off, T, **sampler._kw())
            ctx = (None, h_prev, gates, done_seq, (T, B), need_input_grad, mfma, ("proj", y_bf))
            head_ctx = ([(h_bf, None, dense_chain._shadows(head)[0])], T * B, True)
            samp_ctx = (ms2, ex2, off, None, (T, B, A2))
            out = {"action": None, "log_likelihood": ll.view(T, B)}
            return ctx, h_out, None, h_final, (head_ctx, samp_ctx, out, reg.view(T, B), front_ctx)
        if gi_seq is not None:
            T, B, _ = gi_seq.shape
            x2, pctx = None, "external"
            gi = gi_seq if gi_seq.is_contiguous() else gi_seq.contiguous()
        else:
            T, B, _ = x_seq.shape
            x2 = x_seq.reshape(T * B, self.in_features)
            if not x2.is_contiguous():
                x2 = x2.contiguous()
            if mfma:
                from . import dense_chain

                pctx, gi2 = dense_chain.forward_train([self._proj()], x2, need_input_grad)
                gi = gi2.view(T, B, 3 * H)
            else:
                gi = self._gi(x2).view(T, B, 3 * H)
        if tail is not None:
            from . import dense_chain

            head, sampler, raw_seq = tail
            dense_chain.refresh([head])
            A2 = head.out_features
            ex2 = raw_seq.reshape(T * B, A2 // 2)
            if not ex2.is_contiguous():
                ex2 = ex2.contiguous()
            off = sampler._next_offset()
            h_out, h_prev, gates, h_final, ms2, h_bf, ll, reg = ops.gru_seq_fwd_tail(
                gi, self.w_h.data, self.b_hn.data, state0.contiguous(), done_seq.contiguous(),
                head._ff, head.bias.data, A2, ex2, sampler._state(gi.device), off,
                **sampler._kw())
            ctx = (x2, h_prev, gates, done_seq, (T, B), need_input_grad, mfma, pctx)
            head_ctx = ([(h_bf, None, dense_chain._shadows(head)[0])], T * B, True)
            samp_ctx = (ms2, ex2, off, None, (T, B, A2))
            out = {"action": None, "log_likelihood": ll.view(T, B)}
            return ctx, h_out, None, h_final, (head_ctx, samp_ctx, out, reg.view(T, B))
        h_out, h_prev, gates, h_final = ops.gru_seq_fwd(
            gi, self.w_h.data, self.b_hn.data, state0.contiguous(), done_seq.contiguous(),
            train=True, mfma=mfma)
        ctx = (x2, h_prev, gates, done_seq, (T, B), need_input_grad, mfma, pctx)
        return ctx, h_out, None, h_final

    def replay_backward_tail(self, ctx, head, head_ctx, sampler, samp_ctx, g_out, g_reg):
        """`replay_backward` of a `replay(tail=...)`: the sampler's backward and the head's dX
        run in front of the BPTT inside its launch (`mi_gru_seq_bwd_tail_bf16`); the head's
        dW joins the step's grouped dW requests.  `g_out`: the sampler's output gradient dict.
        Returns the gradient w.r.t. `gi_seq`."""
        x2, h_prev, gates, done_seq, (T, B), need_input_grad, mfma, pctx = ctx
        ms2, ex2, off, eps2, _ = samp_ctx
        H = self.hidden_features
        g_ll = g_out["log_likelihood"]
        g_ll = None if g_ll is None else g_ll.reshape(T * B).contiguous()
        dgi, dgh_bf, dz_bf = ops.gru_seq_bwd_tail(
            gates, h_prev, self.w_h.data, done_seq.contiguous(), head._fb, head.out_features,
            ms2, ex2, sampler._state(ms2.device), off, g_ll, g_reg, eps2=eps2, **sampler._kw())
        ops.dense_bwd_dw_grouped_bf16([(head_ctx[0][0][0], dz_bf, head.kernel.grad,
                                        head.bias.grad)], accumulate=True)
        ops.dense_bwd_dw_grouped_bf16([(h_prev.bf16_image, dgh_bf, self.w_h.grad,
                                        self.b_hn.grad)], accumulate=True, bias_first=[2 * H])
        return dgi

    def replay_backward_proj_tail(self, ctx, head, head_ctx, sampler, samp_ctx, g_out, g_reg):
        """`replay_backward_tail` for a `replay(..., proj=...)`: the projection's backward rides
        in the BPTT launch too.  Queues the dW of the head, of W_h and of W_i; returns dz0_bf
        [T*B, H], the bf16 image of the gradient w.r.t. the pre-activation of the relu layer
        in front (the dz operand of ITS dW, which the caller owns)."""
        _, h_prev, gates, done_seq, (T, B), _, _, (_, y_bf) = ctx
        ms2, ex2, off, eps2, _ = samp_ctx
        H = self.hidden_features
        g_ll = g_out["log_likelihood"]
        g_ll = None if g_ll is None else g_ll.reshape(T * B).contiguous()
        pl = self._proj()
        dgi_bf, dz0_bf, dgh_bf, dz_bf = ops.gru_seq_bwd_proj_tail(
            y_bf, pl._fb, gates, h_prev, self.w_h.data, done_seq.contiguous(), head._fb,
            head.out_features, ms2, ex2, sampler._state(ms2.device), off, g_ll, g_reg, eps2=eps2,
            **sampler._kw())
        ops.dense_bwd_dw_grouped_bf16([(head_ctx[0][0][0], dz_bf, head.kernel.grad,
                                        head.bias.grad)], accumulate=True)
        ops.dense_bwd_dw_grouped_bf16([(h_prev.bf16_image, dgh_bf, self.w_h.grad,
                                        self.b_hn.grad)], accumulate=True, bias_first=[2 * H])
        ops.dense_bwd_dw_grouped_bf16([(y_bf, dgi_bf, self.w_i.grad, self.b_i.grad)],
                                      accumulate=True)
        return dz0_bf

    def replay_backward(self, ctx, g_out, g_reg):
        x2, h_prev, gates, done_seq, (T, B), need_input_grad, mfma, pctx = ctx
        H = self.hidden_features
        dgi, dgh = ops.gru_seq_bwd(g_out.contiguous(), gates, h_prev, self.w_h.data,
                                   done_seq.contiguous(), mfma=mfma, dgh_as_bf16=mfma)
        dgi2 = dgi.view(T * B, 3 * H)
        if mfma:
            # bf16 compute: weight gradients on the bf16 matrix cores too; the sequence
            # kernels left the bf16 images of both operands (no cast launches)
            from . import dense_chain

            hp_bf = getattr(h_prev, "bf16_image", None)
            if hp_bf is None:
                hp_bf = ops.cast_pad_bf16(h_prev.view(T * B, H))
            dgh_bf = dgh if dgh.dtype == torch.bfloat16 else ops.cast_pad_bf16(
                dgh.view(T * B, 3 * H))
            # of the 3H column sums of dgh only the n gate's are a parameter's gradient: they
            # go straight into b_hn.grad (queued with the step's other dW requests, picked
            # out of the slabs by the optimiser launch — no launch of its own, no fill, no add)
            ops.dense_bwd_dw_grouped_bf16([(hp_bf, dgh_bf, self.w_h.grad, self.b_hn.grad)],
                                          accumulate=True, bias_first=[2 * H])
            if isinstance(pctx, str):  # the projection belongs to the caller's chain
                return dgi
            g_x = dense_chain.backward([self._proj()], pctx, dgi2)
            return None if g_x is None else g_x.view(T, B, self.in_features)
        dgh2 = dgh.view(T * B, 3 * H)
        gb_h = torch.zeros(3 * H, dtype=torch.float32, device=dgi.device)
        ops.dense_bwd_dw(h_prev.view(T * B, H), dgh2, None, self.w_h.grad, gb_h, ops.ACT_NONE,
                         accumulate=True)
        self.b_hn.grad += gb_h[2 * H:]
        ops.dense_bwd_dw(x2, dgi2, None, self.w_i.grad, self.b_i.grad, ops.ACT_NONE,
                         accumulate=True)
        if not need_input_grad:
            return None
        g_x = ops.dense_bwd_dx(dgi2, None, self.w_i.data, ops.ACT_NONE)
        return g_x.view(T, B, self.in_features)


def _lstm_act_code(fn, default: int, what: str) -> int:
    """`gate_fn` / `activation_fn` (recurrent.py:36-37) -> kernel code.  Accepted: None
    (the reference's default), the strings / markers of `networks.activations`
    ("sigmoid", "tanh", "relu", "identity" / "none"), and the torch functions of the same
    names.  BPTT keeps the gate OUTPUTS, so only functions whose derivative can be written
    in terms of the output are available (swish is not)."""
    if fn is None:
        return default
    from . import activations as A

    table = {"sigmoid": ops.ACT_SIGMOID, "tanh": ops.ACT_TANH, "relu": ops.ACT_RELU,
             "identity": ops.ACT_NONE, "none": ops.ACT_NONE, "linear": ops.ACT_NONE}
    if isinstance(fn, str) and fn in table:
        return table[fn]
    if isinstance(fn, A._Activation) and fn.name in table:
        return table[fn.name]
    name = getattr(fn, "__name__", None)
    if name in table and fn in (torch.sigmoid, torch.tanh, torch.relu,
                                torch.nn.functional.relu, torch.nn.functional.sigmoid,
                                torch.nn.functional.tanh):
        return table[name]
    raise NotImplementedError(
        f"LSTM: {what}={fn!r} is not available; use 'sigmoid', 'tanh', 'relu' or 'identity' "
        "(functions whose derivative is a function of their output)")


class LSTM(StatefulModule):
    """recurrent.py:16-161 — a wrapper of flax's (Optimized)LSTMCell with the carry
    `(h, c)`, each `[B, H]`: zeros (or, with `trainable_initial_state`, the learnable
    `initial_h` / `initial_c` broadcast over the batch, recurrent.py:85-88,132-161) at
    `initialize_state` and `reset_state`, output = new hidden state,
    `regularization_loss = zeros(B)`, no rollout extras.
    Cell (flax, third-party: arithmetic PARITY UNPINNED), gate order (i, f, g, o),
    bias on the hidden-side projection only:

        a = x W_i + h W_h + b_h
        c' = gate(a_f) c + gate(a_i) act(a_g) ;  h' = gate(a_o) act(c')

    Weights are packed `w_i [in, 4H]`, `w_h [H, 4H]`, `b_h [4H]`.  The input projection
    (with the bias) is one time-batched GEMM; the recurrence runs in a persistent kernel
    (csrc/lstm.hip, H <= 1024; on the bf16 path csrc/lstm_mfma.hip for the default cell
    with H in {32, 64, 96, 128}); BPTT is its mirror plus time-batched GEMMs for the
    weights.  A learnable initial state or non-default `gate_fn` / `activation_fn`
    (sigmoid / tanh / relu / identity) run the fp32 recurrence kernel.  `use_optimized` is
    accepted and ignored (both flax cells compute the same function)."""

    def __init__(self, in_features: int, hidden_features: int, rngs: Rngs, *, gate_fn=None,
                 activation_fn=None, kernel_init=None, recurrent_kernel_init=None,
                 bias_init=None, use_optimized: bool = True,
                 trainable_initial_state: bool = False):
        if hidden_features > 1024:
            raise ValueError("LSTM: hidden_features <= 1024 in this build")
        self.gate_act = _lstm_act_code(gate_fn, ops.ACT_SIGMOID, "gate_fn")
        self.cell_act = _lstm_act_code(activation_fn, ops.ACT_TANH, "activation_fn")
        self.in_features = in_features
        self.hidden_features = hidden_features
        self.trainable_initial_state = bool(trainable_initial_state)
        gen = rngs.generator()
        ki = kernel_init or initializers.lecun_normal()
        kr = recurrent_kernel_init or initializers.orthogonal()
        H = hidden_features
        import numpy as np

        self.w_i = Parameter(np.concatenate([ki(gen, (in_features, H)) for _ in range(4)], axis=1))
        self.w_h = Parameter(np.concatenate([kr(gen, (H, H)) for _ in range(4)], axis=1))
        b = np.zeros(4 * H, dtype=np.float32) if bias_init is None else \
            np.asarray(bias_init(gen, (4 * H,)), dtype=np.float32)
        self.b_h = Parameter(b)
        if self.trainable_initial_state:  # recurrent.py:85-88: single vectors, zeros
            self.initial_h = Parameter(np.zeros(H, dtype=np.float32))
            self.initial_c = Parameter(np.zeros(H, dtype=np.float32))

    def _default_cell(self) -> bool:
        return (not self.trainable_initial_state and self.gate_act == ops.ACT_SIGMOID
                and self.cell_act == ops.ACT_TANH)

    def _mfma(self) -> bool:
        """bf16 compute: the recurrent product runs on the matrix cores (lstm_mfma.hip)."""
        from .. import config

        return (config.compute_dtype() == "bf16" and self._default_cell()
                and ops.gru_mfma_ok(self.hidden_features))

    def _bf16_proj(self) -> bool:
        """bf16 compute: the input projection runs on the bf16 GEMMs."""
        from .. import config

        return config.compute_dtype() == "bf16" and ops.gru_mfma_ok(self.hidden_features)

    def _proj(self) -> _Projection:
        p = self.__dict__.get("_proj_i")
        if p is None or p.kernel is not self.w_i:
            # the hidden-side bias is added with the input projection (a = x W_i + b_h + h W_h)
            p = _Projection(self.w_i, self.b_h, self.in_features, 4 * self.hidden_features)
            self.__dict__["_proj_i"] = p
        return p

    def _gi(self, x2: torch.Tensor) -> torch.Tensor:
        if self._mfma():
            from . import dense_chain

            return dense_chain.forward_infer([self._proj()], x2)
        return ops.dense_fwd(x2, self.w_i.data, self.b_h.data, ops.ACT_NONE)

    def _init_vectors(self):
        if not self.trainable_initial_state:
            return None, None
        return self.initial_h.data, self.initial_c.data

    def _cell_kw(self) -> dict:
        hi, ci = self._init_vectors()
        return dict(h_init=hi, c_init=ci, gate_act=self.gate_act, cell_act=self.cell_act)

    def __call__(self, state, x: torch.Tensor, rollout_extras: Any = None):
        h, c = state
        B = x.shape[0]
        gi = self._gi(x.reshape(B, self.in_features)).view(1, B, 4 * self.hidden_features)
        mfma = self._mfma()
        kw = {} if mfma else self._cell_kw()
        h_out, _, _, _, h_f, c_f = ops.lstm_seq_fwd(gi, self.w_h.data, h.contiguous(),
                                                    c.contiguous(), None, train=False,
                                                    mfma=mfma, **kw)
        return StatefulModuleOutput(next_state=(h_f, c_f), output=h_out[0],
                                    regularization_loss=constant((B,), torch.float32, 0.0, x.device),
                                    metrics={}, rollout_extras=None)

    def initialize_state(self, batch_size: int):
        H = self.hidden_features
        if self.trainable_initial_state:  # recurrent.py:132-141
            return (self.initial_h.data.expand(batch_size, H).contiguous(),
                    self.initial_c.data.expand(batch_size, H).contiguous())
        z = lambda: torch.zeros(batch_size, H, dtype=torch.float32, device=self.device)
        return (z(), z())

    def reset_state(self, prev_state):
        if self.trainable_initial_state:  # recurrent.py:154-158
            return (self.initial_h.data.expand(prev_state[0].shape).contiguous(),
                    self.initial_c.data.expand(prev_state[1].shape).contiguous())
        z = lambda t: constant(t.shape, t.dtype, 0.0, t.device)  # read-only cached zeros
        return (z(prev_state[0]), z(prev_state[1]))

    # ---- training protocol --------------------------------------------------------
    def chain_projection(self):
        """As `GRU.chain_projection`: the input projection for a preceding Dense run's launch."""
        return self._proj() if self._mfma() else None

    def replay(self, state0, x_seq, done_seq, extras_seq, need_input_grad=True, gi_seq=None):
        """`gi_seq` [T, B, 4H]: see `GRU.replay`."""
        H = self.hidden_features
        mfma = self._mfma()
        pctx = None
        if gi_seq is not None:
            T, B, _ = gi_seq.shape
            x2, pctx = None, "external"
            gi = gi_seq if gi_seq.is_contiguous() else gi_seq.contiguous()
        else:
            T, B, _ = x_seq.shape
            x2 = x_seq.reshape(T * B, self.in_features)
            if not x2.is_contiguous():
                x2 = x2.contiguous()
            if mfma:
                from . import dense_chain

                pctx, gi2 = dense_chain.forward_train([self._proj()], x2, need_input_grad)
                gi = gi2.view(T, B, 4 * H)
            else:
                gi = self._gi(x2).view(T, B, 4 * H)
        h0, c0 = state0
        kw = {} if mfma else self._cell_kw()
        h_out, h_prev, c_prev, gates, h_f, c_f = ops.lstm_seq_fwd(
            gi, self.w_h.data, h0.contiguous(), c0.contiguous(), done_seq.contiguous(),
            train=True, mfma=mfma, **kw)
        ctx = (x2, h_prev, c_prev, gates, done_seq, (T, B), need_input_grad, mfma, pctx)
        return ctx, h_out, None, (h_f, c_f)

    def replay_backward(self, ctx, g_out, g_reg):
        x2, h_prev, c_prev, gates, done_seq, (T, B), need_input_grad, mfma, pctx = ctx
        H = self.hidden_features
        if mfma:
            da = ops.lstm_seq_bwd(g_out.contiguous(), gates, c_prev, self.w_h.data,
                                  done_seq.contiguous(), mfma=True)
        elif self.trainable_initial_state:
            # the carry a done row is reset to is (initial_h, initial_c): the BPTT hands
            # back what flows into it.  (The sequence's own first carry is data — the
            # stored pre-rollout state, ppo.py:298-300 — and receives no gradient.)
            da, d_h, d_c = ops.lstm_seq_bwd(
                g_out.contiguous(), gates, c_prev, self.w_h.data, done_seq.contiguous(),
                want_dinit=True, gate_act=self.gate_act, cell_act=self.cell_act)
            self.initial_h.grad.add_(d_h)
            self.initial_c.grad.add_(d_c)
        else:
            da = ops.lstm_seq_bwd(g_out.contiguous(), gates, c_prev, self.w_h.data,
                                  done_seq.contiguous(), gate_act=self.gate_act,
                                  cell_act=self.cell_act)
        da2 = da.view(T * B, 4 * H)
        if mfma:
            from . import dense_chain

            ops.dense_bwd_dw_grouped_bf16(
                [(ops.cast_pad_bf16(h_prev.view(T * B, H)), ops.cast_pad_bf16(da2),
                  self.w_h.grad, None)], accumulate=True)
            if isinstance(pctx, str):  # the projection belongs to the caller's chain
                return da
            g_x = dense_chain.backward([self._proj()], pctx, da2)  # dW_i, db_h, dx
            return None if g_x is None else g_x.view(T, B, self.in_features)
        ops.dense_bwd_dw(h_prev.view(T * B, H), da2, None, self.w_h.grad, self.b_h.grad,
                         ops.ACT_NONE, accumulate=True)
        ops.dense_bwd_dw(x2, da2, None, self.w_i.grad, None, ops.ACT_NONE, accumulate=True)
        if not need_input_grad:
            return None
        g_x = ops.dense_bwd_dx(da2, None, self.w_i.data, ops.ACT_NONE)
        return g_x.view(T, B, self.in_features)
