"""Closed-form batched envs.  All arithmetic is integer hashing plus exact /
single-rounding float ops, so a CPU run and a GPU run of the same env from the
same keys produce bit-identical observations, rewards and flags."""
from __future__ import annotations

from typing import Any

import torch

from .. import random as rnd
from ..algorithms.types import State
from .constants import constant


class MockEnv:
    """`nnx_ppo/test_dummies/mock_env.py:25-63`: observations are unit-variance
    noise regenerated every step from a counter-based key, actions are ignored,
    reward is 1.0, the episode ends when `step_count >= max_steps`.
    `obs_size` may be an int (flat obs) or a dict `{name: size}` (PyTree obs).
    Unlike the reference's mock (whose noise key depends on the step only) each
    env mixes its own key in, so envs are decorrelated."""

    def __init__(self, obs_size, action_size: int, max_steps: int = 5,
                 obs_law: str = "uniform"):
        """`obs_law`: "uniform" (default) — zero-mean unit-variance uniform noise, one IEEE
        multiply on exact operands, so the kernels, the CPU statement and the oracle agree bit
        for bit (what `bench.py` and every parity test use); "normal" — N(0, 1) by Box-Muller,
        the law of the reference's mock (`mock_env.py:43,53`: `jax.random.normal`), evaluated
        with torch ops (no fused env step, no one-launch rollout; CPU / GPU / oracle agree to
        the last ulp of log / cos, not bit for bit).  The kernels' work does not depend on the
        law."""
        if obs_law not in ("uniform", "normal"):
            raise ValueError(f"MockEnv: obs_law must be 'uniform' or 'normal', got {obs_law!r}")
        self.obs_size = obs_size
        self.action_size = action_size
        self.max_steps = max_steps
        self.observation_size = obs_size
        self.obs_law = obs_law

    def _draw(self, key, shape, step):
        if self.obs_law == "normal":
            return rnd.unit_normal(key, shape, fold=step)
        return rnd.unit_uniform(key, shape, fold=step)

    def _obs(self, key: torch.Tensor, step: torch.Tensor) -> Any:
        if isinstance(self.obs_size, dict):
            names = sorted(self.obs_size)
            flat = self._draw(key, (sum(self.obs_size[n] for n in names),), step)
            out, o = {}, 0
            for name in names:
                out[name] = flat[..., o:o + self.obs_size[name]].contiguous()
                o += self.obs_size[name]
            return out
        return self._draw(key, (self.obs_size,), step)

    def reset(self, rng: torch.Tensor) -> State:
        n = rng.shape
        zero = constant(n, torch.int64, 0, rng.device)
        return State(
            data={"key": rng, "step_count": zero},
            obs=self._obs(rng, zero),
            reward=constant(n, torch.float32, 0.0, rng.device),
            done=constant(n, torch.bool, 0, rng.device),
            metrics={}, info={})

    def step(self, state: State, action: torch.Tensor) -> State:
        key = state.data["key"]
        count = state.data["step_count"]
        if (count.is_cuda and count.dim() == 1 and not rnd._TORCH_ONLY[0]
                and self.obs_law == "uniform"):
            # the whole step — counter, done flag and the observation draw — in one launch
            # (csrc/keys.hip: mi_mock_env_step); the same integers and floats as below
            from .. import ops

            if isinstance(self.obs_size, dict):
                names = sorted(self.obs_size)
                step, done, leaves = ops.mock_env_step(key, count, self.max_steps,
                                                       [self.obs_size[n] for n in names])
                obs = dict(zip(names, leaves))
            else:
                step, done, (obs,) = ops.mock_env_step(key, count, self.max_steps,
                                                       [self.obs_size])
        else:
            step = count + 1
            done = step >= self.max_steps
            obs = self._obs(key, step)
        return State(
            data={"key": key, "step_count": step},
            obs=obs,
            reward=constant(step.shape, torch.float32, 1.0, step.device),
            done=done,
            metrics={}, info={})

    def step_deferred(self, state: State, action: torch.Tensor):
        """`step` for a wrapper that can run it inside its own launch
        (`EpisodeWrapper.step_and_reset` -> `mi_mock_episode_step_select`).  Returns
        (state, producer): with a producer the state's `step_count` / `obs` leaves are
        allocated but NOT yet written and `done` is None — the wrapper's launch fills them
        (`producer["leaves"]`: id(tensor) -> (kind, first column)); producer None: an
        ordinary finished step."""
        key = state.data["key"]
        count = state.data["step_count"]
        if not (count.is_cuda and count.dim() == 1 and not rnd._TORCH_ONLY[0]
                and key.is_contiguous() and count.is_contiguous()
                and self.obs_law == "uniform"):
            return self.step(state, action), None
        n, dev = count.shape[0], count.device
        step = torch.empty_like(count)
        leaves = {id(step): (2, 0)}
        keep = [step]
        if isinstance(self.obs_size, dict):
            obs, col = {}, 0
            for name in sorted(self.obs_size):
                w = int(self.obs_size[name])
                obs[name] = torch.empty(n, w, dtype=torch.float32, device=dev)
                leaves[id(obs[name])] = (1, col)
                keep.append(obs[name])
                col += w
        else:
            obs = torch.empty(n, int(self.obs_size), dtype=torch.float32, device=dev)
            leaves[id(obs)] = (1, 0)
            keep.append(obs)
        st = State(data={"key": key, "step_count": step}, obs=obs,
                   reward=constant(step.shape, torch.float32, 1.0, dev), done=None,
                   metrics={}, info={})
        return st, {"key": key, "count": count, "max_steps": int(self.max_steps),
                    "leaves": leaves, "keep": keep}


def cartpole_shaped(max_steps: int = 1000) -> MockEnv:
    """CartpoleBalance-shaped workload: obs 5, action 1 (BASELINE configs 1/2/4/5)."""
    return MockEnv(5, 1, max_steps=max_steps)


def cheetah_shaped(max_steps: int = 1000) -> MockEnv:
    """CheetahRun-shaped workload: PyTree obs {"position": 8, "velocity": 9},
    action 6 (BASELINE config 3)."""
    return MockEnv({"position": 8, "velocity": 9}, 6, max_steps=max_steps)


class DummyCounterEnv:
    """`nnx_ppo/test_dummies/dummy_counter.py:10-43`: reward 1.0 iff the action
    equals the number of steps since the last reset; resets after a random 3..9
    steps.  Used to prove that carry reset stays in lock-step with env reset."""

    observation_size = 1
    action_size = 1

    def reset(self, rng: torch.Tensor) -> State:
        n = rng.shape
        dev = rng.device
        zero = torch.zeros(n, dtype=torch.int64, device=dev)
        return State(
            data={"current_step": zero, "reset_step": rnd.randint(rng, (), 3, 10)},
            obs=torch.zeros(*n, 1, dtype=torch.float32, device=dev),
            info={"current_step": zero},
            reward=torch.ones(n, dtype=torch.float32, device=dev),
            done=torch.zeros(n, dtype=torch.float32, device=dev),
            metrics={})

    def step(self, state: State, action: torch.Tensor) -> State:
        cur = state.data["current_step"] + 1
        data = {"current_step": cur, "reset_step": state.data["reset_step"]}
        done = (cur >= data["reset_step"]).to(torch.float32)
        a = action.reshape(cur.shape)
        return State(
            data=data,
            obs=torch.zeros(*cur.shape, 1, dtype=torch.float32, device=cur.device),
            info={"current_step": cur},
            reward=torch.where(a == cur.to(a.dtype), 1.0, 0.0).to(torch.float32),
            done=done,
            metrics=state.metrics)


class MoveToCenterEnv:
    """`nnx_ppo/test_dummies/move_to_center_env.py:10-50`: 2-D point, reward
    exp(-d^2 / (2 falloff^2)), done when outside `border_radius`."""

    def __init__(self, reward_falloff: float = 0.5, border_radius: float = 2.0):
        self.reward_falloff = reward_falloff
        self.border_radius = border_radius

    def reset(self, rng: torch.Tensor) -> State:
        u = rnd.uniform(rng, (2,))
        phi, rad = u[..., 0], u[..., 1] * (self.border_radius * 0.9)
        ang = 2 * torch.pi * phi
        pos = torch.stack([torch.cos(ang) * rad, torch.sin(ang) * rad], dim=-1)
        return self._get_state({"pos": pos})

    def step(self, state: State, action: torch.Tensor) -> State:
        action = torch.clamp(action, -1, 1)
        return self._get_state({"pos": state.data["pos"] + action})

    def _get_state(self, data) -> State:
        d_sqr = torch.square(data["pos"]).sum(-1)
        reward = torch.exp(-(d_sqr / (self.reward_falloff ** 2) / 2))
        done = torch.where(d_sqr > self.border_radius ** 2, 1.0, 0.0).to(torch.float32)
        return State(data=data, obs=data["pos"] / 10.0, info={}, reward=reward, done=done,
                     metrics={})

    @property
    def observation_size(self):
        return 2

    @property
    def action_size(self):
        return 2


class TwoArmEnv:
    """`nnx_ppo/test_dummies/dict_obs_act_env.py:128-170`: two point masses;
    obs = {"arm1": {"pos", "vel"}, "arm2": {...}} (each `[N, 2]`), action =
    {"arm1": [N, 2], "arm2": [N, 2]}, reward = {"arm1": [N], "arm2": [N]} =
    exp(-|pos|), done when either arm is farther than 3 from the origin.  The PyTree
    reward exercises per-key value heads and `combine_advantages` (ppo.py:440-474)."""

    def _make(self, pos, vel) -> State:
        dist = {k: torch.sqrt(torch.square(p).sum(-1)) for k, p in pos.items()}
        reward = {k: torch.exp(-d) for k, d in dist.items()}
        done = torch.logical_or(dist["arm1"] > 3.0, dist["arm2"] > 3.0)
        obs = {k: {"pos": pos[k], "vel": vel[k]} for k in pos}
        return State(data={}, obs=obs, reward=reward, done=done, metrics={}, info={})

    def reset(self, rng: torch.Tensor) -> State:
        # the reference draws both arms from the same key (dict_obs_act_env.py:139-142)
        p = rnd.uniform(rng, (2,)) * 2.0 - 1.0
        zero = torch.zeros_like(p)
        return self._make({"arm1": p, "arm2": p.clone()}, {"arm1": zero, "arm2": zero.clone()})

    def step(self, state: State, action: dict) -> State:
        vel = {k: state.obs[k]["vel"] + 0.1 * action[k] for k in ("arm1", "arm2")}
        pos = {k: state.obs[k]["pos"] + 0.1 * vel[k] for k in ("arm1", "arm2")}
        return self._make(pos, vel)

    observation_size = {"arm1": {"pos": 2, "vel": 2}, "arm2": {"pos": 2, "vel": 2}}
    action_size = {"arm1": 2, "arm2": 2}
