"""Configuration dataclasses for train_ppo — every field and default of the
reference's `nnx_ppo/algorithms/config.py` (PPOConfig 11-30, EvalConfig 33-42,
VideoConfig 45-57, TrainConfig 60-68, DistillationConfig 71-84, DistillationTrainConfig
87-95, VideoData 98-105, TrainResult 108-116, DistillationTrainResult 119-127)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Optional

import numpy as np

from .types import DistillationState, LoggingLevel, TrainingState


@dataclass
class PPOConfig:
    n_envs: int = 256
    rollout_length: int = 20
    total_steps: int = 512_000
    gae_lambda: float = 0.95
    discounting_factor: float = 0.99
    clip_range: float = 0.2
    learning_rate: float = 1e-4
    normalize_advantages: bool = True
    combine_advantages: bool = False
    n_epochs: int = 4
    n_minibatches: int = 4
    critic_loss_weight: float = 1.0
    gradient_clipping: Optional[float] = None
    weight_decay: Optional[float] = None
    logging_level: LoggingLevel = LoggingLevel.LOSSES
    logging_percentiles: Optional[tuple[int, ...]] = None


@dataclass
class EvalConfig:
    enabled: bool = True
    every_steps: int = 50_000
    n_envs: int = 64
    max_episode_length: int = 1000
    logging_level: LoggingLevel = LoggingLevel.BASIC
    logging_percentiles: Optional[tuple[int, ...]] = (0, 25, 50, 75, 100)


@dataclass
class VideoConfig:
    enabled: bool = False
    every_steps: int = 200_000
    episode_length: int = 1000
    render_kwargs: dict[str, Any] = field(
        default_factory=lambda: {"height": 480, "width": 640}
    )


@dataclass
class BackendConfig:
    """Not in the reference: what `nnx.jit` / XLA decide there is chosen here.
    `compute_dtype` None = the process-wide `nnx_ppo_amd.config.compute_dtype()`;
    `hip_graph` = replay the iteration as one captured HIP graph (the `nnx.jit(ppo_step)`
    of ppo.py:105); `overlap_logging` = enqueue iteration i+1 before the host waits for
    iteration i's metrics when no eval / video / checkpoint is due (algorithms/loop.py)."""
    compute_dtype: Optional[str] = None
    hip_graph: bool = True
    overlap_logging: bool = True


@dataclass
class TrainConfig:
    ppo: PPOConfig = field(default_factory=PPOConfig)
    eval: EvalConfig = field(default_factory=EvalConfig)
    video: VideoConfig = field(default_factory=VideoConfig)
    seed: int = 17
    checkpoint_every_steps: int = 500_000
    backend: BackendConfig = field(default_factory=BackendConfig)


@dataclass
class VideoData:
    frames: np.ndarray  # (T, H, W, C) uint8
    step: int
    episode_reward: float
    episode_length: int


@dataclass
class TrainResult:
    training_state: TrainingState
    final_metrics: dict[str, Any]
    eval_history: list[dict[str, Any]]
    total_steps: int
    total_iterations: int


@dataclass
class DistillationConfig:
    n_envs: int = 256
    rollout_length: int = 20
    total_steps: int = 512_000
    learning_rate: float = 1e-4
    n_epochs: int = 4
    n_minibatches: int = 4
    gradient_clipping: Optional[float] = None
    weight_decay: Optional[float] = None
    logging_level: LoggingLevel = LoggingLevel.LOSSES
    logging_percentiles: Optional[tuple[int, ...]] = None


@dataclass
class DistillationTrainConfig:
    distillation: DistillationConfig = field(default_factory=DistillationConfig)
    eval: EvalConfig = field(default_factory=EvalConfig)
    video: VideoConfig = field(default_factory=VideoConfig)
    seed: int = 17
    checkpoint_every_steps: int = 500_000
    backend: BackendConfig = field(default_factory=BackendConfig)


@dataclass
class DistillationTrainResult:
    training_state: DistillationState
    final_metrics: dict[str, Any]
    eval_history: list[dict[str, Any]]
    total_steps: int
    total_iterations: int
