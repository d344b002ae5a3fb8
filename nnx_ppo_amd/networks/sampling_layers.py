"""Action samplers (counterpart of `nnx_ppo/networks/sampling_layers.py`).

Behaviour is driven by `rollout_extras` exactly as in the reference
(sampling_layers.py:8-22): `None` = ROLLOUT / INFERENCE (sample fresh, emit the
raw action), not-`None` = LOSS_REPLAY (log-likelihood of the stored raw action
under the current policy).  `deterministic` (set by `network.eval()`) returns
the mean instead of sampling.  Output is the sampler dict
`{"action", "log_likelihood"}`; the entropy regulariser `-entropy_weight * H_hat`
is the module's `regularization_loss`.

Noise is Philox, keyed by a device-resident `{seed, offset}` pair: every batch
forward takes the next `offset` (one per call: rollout step, replay, bootstrap),
tracked on the host as `offset_add` and folded into the device counter by
`advance_rng()` at the end of an iteration — so the whole iteration is
HIP-graph capturable and still draws fresh noise on every replay.
"""
from __future__ import annotations

import abc
from typing import Any, Optional

import torch

from .. import ops
from .types import Rngs, StatefulModule, StatefulModuleOutput


class ActionSampler(StatefulModule, abc.ABC):
    deterministic: bool = False


class NormalTanhSampler(ActionSampler):
    """Normal distribution followed by tanh (sampling_layers.py:66-147)."""

    def __init__(self, rng: Rngs, entropy_weight: float, min_std: float = 1e-3,
                 std_scale: float = 1.0):
        self.rng = rng
        self.min_std = min_std
        self.std_scale = std_scale
        self.deterministic = False
        self.entropy_weight = entropy_weight
        self.seed = rng.stream_seed("action_sampling")
        self.rng_state: torch.Tensor | None = None  # {seed, offset} on the device
        self._pending = 0  # calls since the device offset was last advanced
        # test hook: callable(B, A, device) -> (eps, eps2) replacing Philox
        self.noise_override = None

    # -- rng bookkeeping -----------------------------------------------------------
    def _state(self, device) -> torch.Tensor:
        if self.rng_state is None or self.rng_state.device != device:
            off = 0 if self.rng_state is None else int(self.rng_state[1].item())
            self.rng_state = ops.make_rng_state(self.seed, device, off)
        return self.rng_state

    def _next_offset(self) -> int:
        k = self._pending
        self._pending += 1
        return k

    def advance_rng(self) -> None:
        if self._pending and self.rng_state is not None:
            ops.rng_advance(self.rng_state, self._pending)
        self._pending = 0

    def _to_device(self, device) -> None:
        if self.rng_state is not None:
            self.rng_state = self.rng_state.to(device)

    def _kw(self):
        return dict(min_std=self.min_std, std_scale=self.std_scale,
                    entropy_weight=self.entropy_weight)

    def _noise(self, B, A, device):
        if self.noise_override is None:
            return None, None
        return self.noise_override(B, A, device)

    # -- reference interface -----------------------------------------------------------
    def __call__(self, state, mean_and_std: torch.Tensor,
                 rollout_extras: Optional[torch.Tensor] = None) -> StatefulModuleOutput:
        B, A2 = mean_and_std.shape
        eps, eps2 = self._noise(B, A2 // 2, mean_and_std.device)
        r = ops.tanh_gauss_fwd(mean_and_std.contiguous(), rollout_extras,
                               self._state(mean_and_std.device), self._next_offset(),
                               deterministic=self.deterministic, eps=eps, eps2=eps2,
                               want_stats=True, **self._kw())
        return StatefulModuleOutput(
            next_state=(),
            output={"action": r["action"], "log_likelihood": r["log_likelihood"]},
            regularization_loss=r["reg"],
            metrics={"mu": r["mu"], "sigma": r["sigma"]},
            rollout_extras=r["raw"],
        )

    def initialize_state(self, batch_size: int) -> tuple:
        return ()

    # -- training protocol ---------------------------------------------------------------
    def replay(self, state0, x_seq, done_seq, extras_seq, need_input_grad=True):
        T, B, A2 = x_seq.shape
        ms2 = x_seq.reshape(T * B, A2)
        if not ms2.is_contiguous():  # a slice handed over by a routing container
            ms2 = ms2.contiguous()
        ex2 = extras_seq.reshape(T * B, A2 // 2)
        if not ex2.is_contiguous():
            ex2 = ex2.contiguous()
        off = self._next_offset()
        eps, eps2 = self._noise(T * B, A2 // 2, x_seq.device)
        r = ops.tanh_gauss_fwd(ms2, ex2, self._state(x_seq.device), off,
                               deterministic=self.deterministic, eps=eps, eps2=eps2,
                               want_action=False, want_raw=False, **self._kw())
        out = {"action": None, "log_likelihood": r["log_likelihood"].view(T, B)}
        return (ms2, ex2, off, eps2, (T, B, A2)), out, r["reg"].view(T, B), ()

    def replay_backward(self, ctx, g_out, g_reg):
        ms2, ex2, off, eps2, (T, B, A2) = ctx
        g_ll = g_out["log_likelihood"]
        g_ll = None if g_ll is None else g_ll.reshape(T * B)
        g = ops.tanh_gauss_bwd(ms2, ex2, self._state(ms2.device), off, g_ll, g_reg, eps2=eps2,
                               **self._kw())
        return g.view(T, B, A2)
