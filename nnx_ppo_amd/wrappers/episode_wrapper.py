"""EpisodeWrapper (counterpart of `nnx_ppo/wrappers/episode_wrapper.py:7-39`),
batched: `info["step_counter"]` is int64 `[n_envs]`, `info["truncated"]` bool
`[n_envs]`, `done` is returned as float like the reference (`.astype(float)`).
Counters and flags are integer / boolean torch ops — bit-exact on any device."""
from __future__ import annotations

import torch

from .. import random as rnd
from ..envs.constants import cast_constant, constant, is_constant


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
