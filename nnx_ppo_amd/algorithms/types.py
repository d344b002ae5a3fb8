"""Runtime types of the PPO path (counterpart of `nnx_ppo/algorithms/types.py`).

Same names and field order as the reference: `EnvState` / `RLEnv` protocols
(types.py:15-45), `TrainingState` (48-56), `Transition` (59-80), `LoggingLevel`
(128-150).  Leaves are torch tensors; containers are pytrees (`nnx_ppo_amd.tree`).

Env convention (the one deliberate difference, SURVEY §7 hard part 5): the
reference writes single-env `reset/step` and `jax.vmap`s them
(`rollout.py:21,39`); torch has no tracing vmap over arbitrary Python, so envs
here are *batched*: every leaf of the state carries a leading `n_envs` axis,
`reset(rng)` takes an int64 key tensor `[n_envs]` and `step(state, action)`
takes batched actions.
"""
from __future__ import annotations

import dataclasses
import enum
from typing import Any, Protocol, runtime_checkable

import torch

from ..networks.types import PPONetworkOutput
from ..tree import TreeDataclass


@runtime_checkable
class EnvState(Protocol):
    """types.py:15-34 — obs / done / reward / info / metrics (+ replace at runtime)."""

    @property
    def obs(self) -> Any: ...
    @property
    def done(self) -> torch.Tensor: ...  # bool or float depending on env
    @property
    def reward(self) -> Any: ...
    @property
    def info(self) -> dict[str, Any]: ...
    @property
    def metrics(self) -> dict[str, Any]: ...


@runtime_checkable
class RLEnv(Protocol):
    """types.py:37-45."""

    def reset(self, rng: torch.Tensor) -> EnvState: ...
    def step(self, state: Any, action: Any) -> EnvState: ...


@dataclasses.dataclass(frozen=True)
class State(TreeDataclass):
    """Concrete env state with the fields of `mujoco_playground.State` that the
    reference's envs and wrappers use (`test_dummies/dummy_counter.py:16-26`,
    `wrappers/episode_wrapper.py:14-31`)."""

    data: Any
    obs: Any
    reward: Any
    done: torch.Tensor
    metrics: dict = dataclasses.field(default_factory=dict)
    info: dict = dataclasses.field(default_factory=dict)


@dataclasses.dataclass(frozen=True)
class TrainingState(TreeDataclass):
    """types.py:48-56.  `steps_taken` is an int64 device scalar (the reference
    keeps an fp32 scalar, ppo.py:571, which stops counting exactly at 2^24)."""

    networks: Any
    network_states: Any
    env_states: Any
    optimizer: Any
    rng_key: torch.Tensor
    steps_taken: torch.Tensor


@dataclasses.dataclass(frozen=True)
class Transition(TreeDataclass):
    """types.py:59-80 — one rollout, time-major `[T, N, ...]` leaves."""

    obs: Any
    network_output: PPONetworkOutput
    rewards: Any
    done: torch.Tensor
    truncated: torch.Tensor
    next_obs: Any
    metrics: dict
    rollout_extras: Any = None


@dataclasses.dataclass(frozen=True)
class DistillationTransition(TreeDataclass):
    """types.py:85-106 — one distillation rollout: the student's outputs drive the env,
    the teacher's `rollout_extras` (teacher in eval mode: its action mean at the sampler
    positions) are the target."""

    obs: Any
    student_output: PPONetworkOutput
    rewards: Any
    done: torch.Tensor
    truncated: torch.Tensor
    next_obs: Any
    metrics: dict
    student_rollout_extras: Any = None
    teacher_rollout_extras: Any = None


@dataclasses.dataclass(frozen=True)
class DistillationState(TreeDataclass):
    """types.py:111-125.  The teacher is an external argument (like the env); only its
    per-env carry is tracked here."""

    student: Any
    student_states: Any
    teacher_states: Any
    env_states: Any
    optimizer: Any
    rng_key: torch.Tensor
    steps_taken: torch.Tensor


class LoggingLevel(enum.Flag):
    """types.py:128-150."""

    LOSSES = enum.auto()
    CRITIC_EXTRA = enum.auto()
    ACTOR_EXTRA = enum.auto()
    TRAIN_ROLLOUT_STATS = enum.auto()
    ROLLOUT_OBS = enum.auto()
    TRAINING_ENV_METRICS = enum.auto()
    GRAD_NORM = enum.auto()
    WEIGHTS = enum.auto()
    THROUGHPUT = enum.auto()
    BASIC = LOSSES
    ALL = (
        LOSSES
        | ACTOR_EXTRA
        | CRITIC_EXTRA
        | TRAIN_ROLLOUT_STATS
        | TRAINING_ENV_METRICS
        | GRAD_NORM
        | WEIGHTS
        | ROLLOUT_OBS
        | THROUGHPUT
    )
    NONE = 0
