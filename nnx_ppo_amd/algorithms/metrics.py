"""Metric reductions (counterpart of `nnx_ppo/algorithms/metrics.py:17-121`).
Keys and reduction rules follow the code, not the (stale) docs: nested mappings
join with '/', bool arrays log their mean, everything else logs mean/std or the
requested percentiles.  Runs on tiny tensors (one value per gradient step)."""
from __future__ import annotations

from collections.abc import Mapping
from typing import Any, Optional

import torch

from .types import LoggingLevel, Transition


_PCT_CACHE: dict = {}


def percentiles(x: torch.Tensor, levels: tuple) -> torch.Tensor:
    """`jp.percentile(x, levels)` (linear interpolation between order statistics) from one
    sort.  The interpolation indices and weights depend on (levels, x.numel()) only, so
    they are built once per device and cached: nothing is copied from the host when the
    call is replayed inside a captured HIP graph, and — unlike `torch.quantile` — there is
    no 16M-element limit."""
    flat = x.float().reshape(-1)
    n = flat.numel()
    key = (tuple(float(l) for l in levels), n, str(flat.device))
    c = _PCT_CACHE.get(key)
    if c is None:
        pos = [min(max(float(l), 0.0), 100.0) / 100.0 * (n - 1) for l in levels]
        lo = [int(p) for p in pos]
        hi = [min(i + 1, n - 1) for i in lo]
        w = [p - i for p, i in zip(pos, lo)]
        c = (torch.tensor(lo, dtype=torch.int64, device=flat.device),
             torch.tensor(hi, dtype=torch.int64, device=flat.device),
             torch.tensor(w, dtype=torch.float32, device=flat.device))
        _PCT_CACHE[key] = c
    lo, hi, w = c
    s = flat.sort().values
    a, b = s[lo], s[hi]
    return a + (b - a) * w


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
        pct = percentiles(x, tuple(percentile_levels))
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
