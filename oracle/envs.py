"""Batched CPU restatement of the synthetic env and of the episode wrapper.

TEST INFRASTRUCTURE ONLY (see `oracle/__init__.py`).  Written from the reference files, with
none of the product's code (keys come from `oracle.keys`):

  * `MockEnv` — `nnx_ppo/test_dummies/mock_env.py:25-63`: actions ignored, reward 1.0 per
    step (0.0 at reset), `step_count` += 1, done when `step_count >= max_steps`,
    observations = fresh noise every step.  Two declared differences from the reference
    file, both the build's: the noise of env e at step s is drawn from
    fold_key(key_e, s) — the reference's key depends on the step only
    (`mock_env.py:54`: `PRNGKey(step_count + 1)`), which would make every env see the same
    observation — and it is zero-mean unit-variance UNIFORM, (u − ½)√12, where the reference
    draws `jax.random.normal` (threefry: parity unpinned either way).  A dict `obs_size` is a
    PyTree observation: the flat draw cut at the leaf widths, names sorted.
  * `EpisodeWrapper` — `nnx_ppo/wrappers/episode_wrapper.py:8-44`, line for line in the
    vmapped (batched) form the rollout sees (`rollout.py:21,39`).

States are dataclasses so that the oracle's `tree_where` (rollout.py:270-279) descends them.
"""
from __future__ import annotations

import dataclasses
from typing import Any

import torch

from . import keys as K


@dataclasses.dataclass
class State:
    data: dict
    obs: Any
    reward: torch.Tensor
    done: torch.Tensor
    metrics: dict
    info: dict

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)


class MockEnv:
    def __init__(self, obs_size, action_size: int, max_steps: int = 5, obs_law: str = "uniform"):
        self.obs_size, self.action_size, self.max_steps = obs_size, action_size, max_steps
        self.observation_size = obs_size
        self.obs_law = obs_law  # "normal": the reference's own law (mock_env.py:43,53)

    def _draw(self, key, shape, step):
        return (K.unit_normal if self.obs_law == "normal" else K.unit_uniform)(key, shape,
                                                                               fold=step)

    def _obs(self, key: torch.Tensor, step: torch.Tensor):
        if isinstance(self.obs_size, dict):
            names = sorted(self.obs_size)
            flat = self._draw(key, (sum(self.obs_size[n] for n in names),), step)
            out, o = {}, 0
            for n in names:
                out[n] = flat[..., o:o + self.obs_size[n]].contiguous()
                o += self.obs_size[n]
            return out
        return self._draw(key, (self.obs_size,), step)

    def reset(self, rng: torch.Tensor) -> State:  # mock_env.py:40-50
        zero = torch.zeros(rng.shape, dtype=torch.int64)
        return State(data={"key": rng, "step_count": zero}, obs=self._obs(rng, zero),
                     reward=torch.zeros(rng.shape, dtype=torch.float32),
                     done=torch.zeros(rng.shape, dtype=torch.bool), metrics={}, info={})

    def step(self, state: State, action) -> State:  # mock_env.py:52-63
        key = state.data["key"]
        step = state.data["step_count"] + 1
        return State(data={"key": key, "step_count": step}, obs=self._obs(key, step),
                     reward=torch.ones(step.shape, dtype=torch.float32),
                     done=step >= self.max_steps, metrics={}, info={})


class EpisodeWrapper:
    def __init__(self, env, max_len: int):  # episode_wrapper.py:8-11
        self.env, self.max_len = env, max_len

    def step(self, state: State, action) -> State:  # episode_wrapper.py:13-24
        nxt = self.env.step(state, action)
        info = dict(nxt.info)
        info["step_counter"] = state.info["step_counter"] + 1
        over = info["step_counter"] >= self.max_len
        prev = nxt.info.get("truncated", None)
        truncated = over if prev is None else torch.logical_or(prev, over)
        info["truncated"] = truncated
        done = torch.logical_or(nxt.done.to(torch.bool), truncated).to(torch.float32)
        return nxt.replace(done=done, info=info)

    def reset(self, rng: torch.Tensor) -> State:  # episode_wrapper.py:26-34
        ks = K.split(rng)
        base_rng, step_counter_rng = ks[..., 0].contiguous(), ks[..., 1].contiguous()
        nxt = self.env.reset(base_rng)
        info = dict(nxt.info)
        info["step_counter"] = K.randint(step_counter_rng, (), 0, self.max_len // 2)
        info["truncated"] = torch.zeros(rng.shape, dtype=torch.bool)
        # the product keeps `done` float from the reset on (what `step` returns,
        # episode_wrapper.py:21), so that carries have stable leaf dtypes
        return nxt.replace(done=nxt.done.to(torch.float32), info=info)

    @property
    def observation_size(self):
        return self.env.observation_size

    @property
    def action_size(self):
        return self.env.action_size
