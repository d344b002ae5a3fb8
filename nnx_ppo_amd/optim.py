"""Flat-arena optimiser in the role of `nnx.Optimizer(networks,
optax.chain([clip_by_global_norm?], adam | adamw), wrt=nnx.Param)`
(`nnx_ppo/algorithms/ppo.py:555-569`).

All parameters of a network are re-homed into ONE contiguous fp32 arena (and
their gradients / Adam moments into three more), each parameter starting on a
256-byte boundary, so that
  * the whole optimiser step is one kernel launch (`mi_adam_step_f32`),
  * the multi-GPU gradient exchange is one all-reduce of one buffer,
  * zeroing gradients + advancing the step counter is one launch.
`Parameter.data` / `.grad` become views of the arenas; modules keep working
unchanged.  The step counter and gradient norm stay on the device."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import ops, parallel
from .networks.types import StatefulModule, bump_param_epoch

_ALIGN = 64  # floats (256 B)
# MIPPO_DEFER_DW=0: always reduce the dW slabs in their own launch (A/B switch)
DEFER_DW = os.environ.get("MIPPO_DEFER_DW", "1") != "0"


class Optimizer:
    def __init__(self, networks: StatefulModule, learning_rate: float = 1e-4,
                 gradient_clipping: Optional[float] = None,
                 weight_decay=None, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8,
                 device=None):
        self.learning_rate = float(learning_rate)
        self.gradient_clipping = gradient_clipping
        # ppo.py:558-566: None -> adam; True -> adamw with optax's default 1e-4;
        # a float -> adamw with that decay.
        if weight_decay is None or weight_decay is False:
            self.weight_decay = 0.0
        elif weight_decay is True:
            self.weight_decay = 1e-4
        else:
            self.weight_decay = float(weight_decay)
        self.b1, self.b2, self.eps = b1, b2, eps
        named = networks.named_parameters()
        device = torch.device(device) if device is not None else networks.device
        self.device = device
        offsets = []
        total = 0
        for _, p in named:
            offsets.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.n = max(total, _ALIGN)
        self.names = [n for n, _ in named]
        self.offsets = offsets
        self.shapes = [tuple(p.shape) for _, p in named]
        self.params = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.grads = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.m = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.v = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.step = torch.zeros(1, dtype=torch.int64, device=device)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=device)
        self._clean = True  # the gradient arena is all zeros
        for (_, p), off in zip(named, offsets):
            n = p.numel()
            self.params[off:off + n].copy_(p.data.reshape(-1).to(device))
            p.data = self.params[off:off + n].view(p.shape)
            p.grad = self.grads[off:off + n].view(p.shape)
        bump_param_epoch()

    # ---- one gradient step = begin() ... backward ... update() --------------------
    def begin(self, defer_dw: bool = False) -> None:
        """Start a gradient step with a zeroed gradient arena.  `update()` leaves the
        arena zeroed (and counts the step) in the Adam launch itself, so in the
        training loop this is free; it only launches when gradients were accumulated
        since the last update (e.g. two loss evaluations without an update).

        `defer_dw`: the caller promises to read the gradients only through this
        optimiser until `update()`; the split-M slabs of the step's last grouped dW
        launch then stay unreduced and `update()` sums them inside the Adam launch (same
        summation order, one launch fewer).  Without it `.grad` is complete after every
        backward."""
        ops.flush_pending_slabs()
        if not self._clean:
            ops.begin_grad_step(self.grads, None)
        self._clean = False
        ops.slab_defer.arena = self.grads if (defer_dw and DEFER_DW) else None

    def _pending_slabs(self):
        """The pending dW slabs in `adam_step`'s form (None: nothing pending)."""
        pend = ops.take_pending_slabs()
        if pend is None:
            return None
        sp, S, Ks, Ns, _, _, gw_off, gb_off, _, b_lo = pend
        return (sp, S, Ks, Ns, gw_off, gb_off, b_lo)

    def compute_grad_norm(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        ops.flush_pending_slabs()
        return ops.global_norm(self.grads, out=self.grad_norm if out is None else out)

    def update(self, have_norm: bool = False, norm_out: Optional[torch.Tensor] = None) -> None:
        """All-reduce (if sharded), optional global-norm clip, Adam(W).  `norm_out`
        (float32 [1]) receives the global norm of the gradient that is applied — in a
        sharded run the norm AFTER the all-reduce, the one `clip_by_global_norm` sees
        (`ppo.py:313-316`).  `have_norm`: `self.grad_norm` already holds it (single GPU)."""
        from .networks import dense_chain

        shadowed = dense_chain.shadows_in(self.params)[:16]
        table = [(off, l.kernel.shape[0], l.kernel.shape[1], l._w_bf, l._wt_bf, l._ff, l._fb)
                 for l, off in shadowed]
        ops.slab_defer.arena = None
        comm = parallel.peer_comm() if parallel.is_distributed() else None
        # the one-shot exchange fused into Adam sums the pending slabs itself (per chunk, before
        # the push); every other reader of the whole gradient needs them reduced first
        fused_exchange = (comm is not None and norm_out is None
                          and self.gradient_clipping is None
                          and self.n * 4 <= comm.slot_bytes)
        if (parallel.is_distributed() and not fused_exchange) or norm_out is not None or \
                (self.gradient_clipping is not None and not have_norm):
            ops.flush_pending_slabs()  # somebody reads the whole gradient before Adam
        if parallel.is_distributed():
            if fused_exchange and comm.adam_step_allreduce(
                    self.params, self.grads, self.m, self.v, self.step,
                    lr=self.learning_rate, b1=self.b1, b2=self.b2, eps=self.eps,
                    weight_decay=self.weight_decay, shadows=table,
                    slabs=self._pending_slabs()):
                # slab sums + exchange + mean + Adam + bf16 images + gradient zeroing: one launch
                self._clean = True
                bump_param_epoch()
                dense_chain.mark_fresh([l for l, _ in shadowed])
                return
            ops.flush_pending_slabs()
            parallel.allreduce_mean_(self.grads)
            have_norm = False
        gn = None
        if norm_out is not None:
            gn = self.compute_grad_norm(norm_out)
        elif self.gradient_clipping is not None:
            gn = self.grad_norm if have_norm else self.compute_grad_norm()
        if self.gradient_clipping is None:
            gn = None  # logged only: the Adam launch must not clip
        # Dense kernels with bf16 shadows get the new values written into the shadows by
        # the same launch (up to 16 layers; the rest refresh lazily at their next use)
        ops.adam_step(self.params, self.grads, self.m, self.v, self.step,
                      lr=self.learning_rate, b1=self.b1, b2=self.b2, eps=self.eps,
                      weight_decay=self.weight_decay, grad_norm=gn,
                      max_norm=float(self.gradient_clipping or 0.0), begin_next=True,
                      shadows=table, slabs=self._pending_slabs())
        self._clean = True
        bump_param_epoch()
        dense_chain.mark_fresh([l for l, _ in shadowed])

    # ---- state export (checkpoint callbacks) ----------------------------------------
    def state_dict(self) -> dict:
        return {"params": self.params.clone(), "m": self.m.clone(), "v": self.v.clone(),
                "step": self.step.clone(), "names": list(self.names),
                "offsets": list(self.offsets), "shapes": list(self.shapes)}

    def load_state_dict(self, sd: dict) -> None:
        self.params.copy_(sd["params"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step.copy_(sd["step"])
        self._clean = False
        bump_param_epoch()
