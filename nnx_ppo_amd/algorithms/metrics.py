"""Metric reductions (counterpart of `nnx_ppo/algorithms/metrics.py:17-121`).
Keys and reduction rules follow the code, not the (stale) docs: nested mappings
join with '/', bool arrays log their mean, everything else logs mean/std or the
requested percentiles.  Runs on tiny tensors (one value per gradient step)."""
from __future__ import annotations

from collections.abc import Mapping
from typing import Any, Optional

import torch

from .types import LoggingLevel, Transition


def _log_metric(metrics: dict, name: str, x: Any,
                percentile_levels: Optional[tuple] = None) -> None:
    """metrics.py:72-100."""
    if isinstance(x, Mapping):
        for k, v in x.items():
            _log_metric(metrics, f"{name}/{k}", v, percentile_levels)
        return
    if x is None:
        return
    if x.dtype == torch.bool:
        metrics[name] = x.float().mean()
    elif percentile_levels is None or len(percentile_levels) == 0:
        # one Welford launch for both (population std, as `jp.std`)
        sd, mu = torch.std_mean(x.float(), correction=0)
        metrics[f"{name}/mean"] = mu
        metrics[f"{name}/std"] = sd
    else:
        q = torch.tensor(percentile_levels, dtype=torch.float32, device=x.device) / 100.0
        pct = torch.quantile(x.float().reshape(-1), q)
        for pl, p in zip(percentile_levels, pct):
            metrics[f"{name}/p{int(pl)}"] = p


def compute_metrics(loss_metrics: dict, rollout_data: Transition, logging_level: LoggingLevel,
                    percentile_levels: Optional[tuple] = None) -> dict:
    """metrics.py:17-69."""
    metrics: dict = {}
    for k, v in loss_metrics.items():
        _log_metric(metrics, k, v, percentile_levels)
    if LoggingLevel.TRAINING_ENV_METRICS in logging_level:
        for k, v in rollout_data.metrics.items():
            _log_metric(metrics, k, v, percentile_levels)
    if LoggingLevel.TRAIN_ROLLOUT_STATS in logging_level:
        _log_metric(metrics, "rollout_batch/reward", rollout_data.rewards, percentile_levels)
        _log_metric(metrics, "rollout_batch/action", rollout_data.network_output.actions,
                    percentile_levels)
        metrics["rollout_batch/done_rate"] = rollout_data.done.float().mean()
        metrics["rollout_batch/truncation_rate"] = rollout_data.truncated.float().mean()
    if LoggingLevel.ACTOR_EXTRA in logging_level:
        _log_metric(metrics, "loglikelihood", rollout_data.network_output.loglikelihoods,
                    percentile_levels)
    if LoggingLevel.CRITIC_EXTRA in logging_level:
        _log_metric(metrics, "losses/predicted_value",
                    rollout_data.network_output.value_estimates, percentile_levels)
    return metrics


def log_weight_stats(metrics: dict, flat_params: torch.Tensor,
                     percentile_levels: Optional[tuple] = None) -> None:
    """metrics.py:103-121 (over the flat parameter arena)."""
    _log_metric(metrics, "weights", flat_params, percentile_levels)
