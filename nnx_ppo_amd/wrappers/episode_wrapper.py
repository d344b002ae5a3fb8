"""EpisodeWrapper (counterpart of `nnx_ppo/wrappers/episode_wrapper.py:7-39`),
batched: `info["step_counter"]` is int64 `[n_envs]`, `info["truncated"]` bool
`[n_envs]`, `done` is returned as float like the reference (`.astype(float)`).
Counters and flags are integer / boolean torch ops — bit-exact on any device."""
from __future__ import annotations

import torch

from .. import random as rnd
from ..envs.constants import cast_constant, constant, is_constant


class _Produced:
    """Marks a state leaf that `step` itself produces (fused step + reset select)."""

    def __init__(self, name: str):
        self.name = name

    def __repr__(self):
        return f"<{self.name}>"


_COUNTER, _TRUNC, _DONE = _Produced("step_counter"), _Produced("truncated"), _Produced("done")


class EpisodeWrapper:
    def __init__(self, env, max_len: int):
        self.env = env
        self.max_len = max_len

    def step(self, state, action):
        next_state = self.env.step(state, action)
        if state.info["step_counter"].is_cuda:
            # the same integer / flag arithmetic in one launch (csrc/keys.hip)
            from .. import ops

            prev = next_state.info.get("truncated", None)
            c, t, d, flag = ops.episode_step(state.info["step_counter"], next_state.done,
                                             prev if isinstance(prev, torch.Tensor) else None,
                                             self.max_len)
            d.done_flag = flag  # the same flag as bool: the rollout's reset select reads it
            next_state.info["step_counter"] = c
            next_state.info["truncated"] = t
            return next_state.replace(done=d)
        next_state.info["step_counter"] = state.info["step_counter"] + 1
        prev_trunc = next_state.info.get("truncated", False)
        over = next_state.info["step_counter"] >= self.max_len
        truncated = over if prev_trunc is False else torch.logical_or(prev_trunc, over)
        next_state.info["truncated"] = truncated
        done = torch.logical_or(next_state.done.to(torch.bool), truncated)
        return next_state.replace(done=done.to(torch.float32))

    def step_and_reset(self, state, action, reset_states):
        """`step` followed by the rollout's `tree_where(done, reset_states, stepped)`
        (rollout.py:41-44) — on the GPU both in ONE launch (`mi_episode_step_select`).
        Returns (stepped state, state after the reset select); the second is None when
        the fused launch does not apply and the caller selects itself."""
        counter = state.info["step_counter"]
        r_info = getattr(reset_states, "info", None) or {}
        rc, rt, rd = r_info.get("step_counter"), r_info.get("truncated"), reset_states.done
        ok = (counter.is_cuda and counter.dim() == 1
              and all(isinstance(x, torch.Tensor) and x.shape == counter.shape
                      and x.is_contiguous() for x in (rc, rt, rd))
              and rc.dtype == torch.int64 and rt.dtype == torch.bool
              and rd.dtype == torch.float32)
        if not ok:
            return self.step(state, action), None
        from .. import ops
        from ..algorithms.rollout import collect_pairs, fill_slots

        # an env that can hand its own step to this launch (MockEnv.step_deferred): no
        # launch of its own
        deferred = getattr(self.env, "step_deferred", None)
        inner, producer = deferred(state, action) if deferred is not None \
            else (self.env.step(state, action), None)
        prev = inner.info.get("truncated", None)
        prev = prev if isinstance(prev, torch.Tensor) else None
        marked_info = dict(inner.info)
        marked_info["step_counter"] = _COUNTER
        marked_info["truncated"] = _TRUNC
        marked = inner.replace(done=_DONE, info=marked_info)
        pairs, skeleton = collect_pairs(
            counter.shape[0], reset_states, marked,
            special=lambda x, y: y if isinstance(y, _Produced) else None)
        if producer is not None and (len(pairs) > 16 or not all(
                any(y is leaf for _, y in pairs) for leaf in producer["keep"])):
            # the fused launch does not apply after all: the env steps by itself
            inner, producer = self.env.step(state, action), None
            marked = inner.replace(done=_DONE, info=marked_info)
            pairs, skeleton = collect_pairs(
                counter.shape[0], reset_states, marked,
                special=lambda x, y: y if isinstance(y, _Produced) else None)
        if len(pairs) > 16:
            return self._finish_step(inner, state, prev), None
        c, t, d, flag, c_sel, t_sel, d_sel, outs = ops.episode_step_select(
            counter, inner.done, prev, self.max_len, rc, rt, rd, pairs, producer=producer)
        d.done_flag = flag
        info = dict(inner.info)
        info["step_counter"] = c
        info["truncated"] = t
        stepped = inner.replace(done=d, info=info)
        sel = {_COUNTER: c_sel, _TRUNC: t_sel, _DONE: d_sel}
        after = fill_slots(skeleton, outs,
                           special=lambda v: sel[v] if isinstance(v, _Produced) else v)
        return stepped, after

    def _finish_step(self, next_state, state, prev):
        from .. import ops

        c, t, d, flag = ops.episode_step(state.info["step_counter"], next_state.done, prev,
                                         self.max_len)
        d.done_flag = flag
        next_state.info["step_counter"] = c
        next_state.info["truncated"] = t
        return next_state.replace(done=d)

    def reset(self, rng: torch.Tensor):
        base_rng, step_counter_rng = rnd.split2(rng)
        next_state = self.env.reset(base_rng)
        next_state.info["step_counter"] = rnd.randint(step_counter_rng, (), 0, self.max_len // 2)
        next_state.info["truncated"] = constant(rng.shape, torch.bool, 0, rng.device)
        # `step` returns done as float (episode_wrapper.py:21): keep the reset state's
        # leaf the same dtype so rollout carries (and captured graphs) see stable leaves
        done = next_state.done
        if done.dtype != torch.float32:
            done = cast_constant(done, torch.float32) if is_constant(done) \
                else done.to(torch.float32)
        return next_state.replace(done=done)

    @property
    def observation_size(self):
        return self.env.observation_size

    @property
    def action_size(self):
        return self.env.action_size
