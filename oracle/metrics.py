"""Metric reductions and the evaluation rollout of the oracle (numpy / torch-CPU).
TEST INFRASTRUCTURE — see oracle/__init__.py.

Restated from the reference:
  _log_metric / compute_metrics        nnx_ppo/algorithms/metrics.py:17-100
  _add_reward_metrics / eval_rollout   nnx_ppo/algorithms/rollout.py:76-148
  grad-norm / CRITIC_EXTRA rows        nnx_ppo/algorithms/ppo.py:313-315,509-528

Python loops over time, numpy reductions: written for the small cases the parity
tests use.  `jp.std` is the population standard deviation; `jp.percentile` interpolates
linearly between order statistics (numpy's default, which is what is called here).
"""
from __future__ import annotations

from collections.abc import Mapping
from typing import Any, Optional

import numpy as np
import torch

from .ppo import _as_bool, _tmap


def _np(x) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().double().numpy()
    return np.asarray(x, dtype=np.float64)


def log_metric(metrics: dict, name: str, x: Any, percentile_levels: Optional[tuple] = None):
    """metrics.py:72-100."""
    if isinstance(x, Mapping):
        for k, v in x.items():
            log_metric(metrics, f"{name}/{k}", v, percentile_levels)
        return
    if x is None:
        return
    if isinstance(x, torch.Tensor) and x.dtype == torch.bool:
        metrics[name] = float(_np(x.to(torch.float64)).mean())
    elif not percentile_levels:
        a = _np(x)
        metrics[f"{name}/mean"] = float(a.mean())
        metrics[f"{name}/std"] = float(a.std())
    else:
        pct = np.percentile(_np(x).reshape(-1), np.asarray(percentile_levels, dtype=np.float64))
        for pl, p in zip(percentile_levels, pct):
            metrics[f"{name}/p{int(pl)}"] = float(p)


def add_reward_metrics(out: dict, name: str, reward: Any, percentile_levels: Optional[tuple]):
    """rollout.py:76-94."""
    if isinstance(reward, Mapping):
        for k, v in reward.items():
            add_reward_metrics(out, f"{name}/{k}", v, percentile_levels)
    elif percentile_levels is not None:
        pct = np.percentile(_np(reward), np.asarray(percentile_levels, dtype=np.float64))
        for pl, p in zip(percentile_levels, pct):
            out[f"{name}/p{int(pl)}"] = float(p)
    else:
        a = _np(reward)
        out[f"{name}/mean"] = float(a.mean())
        out[f"{name}/std"] = float(a.std())


def eval_rollout(env, networks, n_envs: int, max_episode_length: int, key, keys,
                 logging_percentiles: Optional[tuple] = None) -> dict:
    """rollout.py:97-148 — the scan written as a loop.  `networks` is an oracle module in
    eval mode (deterministic samplers); `keys` provides the integer key plumbing."""
    env_state = env.reset(keys.split(key, n_envs))
    net_state = networks.initialize_state(n_envs)
    cuml = _tmap(lambda r: torch.zeros_like(r, dtype=torch.float64), env_state.reward)
    lifespan = torch.zeros(n_envs, dtype=torch.float64)
    for _ in range(max_episode_length):
        with torch.no_grad():
            out = networks(net_state, env_state.obs)
        net_state = out.next_state
        nxt = env.step(env_state, _tmap(lambda a: a.to(torch.float32), out.output.actions))
        prev_done = _as_bool(env_state.done)
        sticky = torch.logical_or(_as_bool(nxt.done), prev_done)      # rollout.py:114-116
        nxt = nxt.replace(done=sticky.to(torch.float32))
        this = _tmap(lambda r: torch.where(prev_done, torch.zeros_like(r), r).double(),
                     nxt.reward)                                       # rollout.py:118-121
        cuml = _tmap(torch.add, cuml, this)
        lifespan = lifespan + torch.where(sticky, 0.0, 1.0)            # rollout.py:123
        env_state = nxt
    a = _np(lifespan)
    metrics = dict(lifespan_mean=float(a.mean()), lifespan_std=float(a.std()))
    add_reward_metrics(metrics, "episode_reward", cuml, logging_percentiles)
    if logging_percentiles is not None:
        pct = np.percentile(a, np.asarray(logging_percentiles, dtype=np.float64))
        for pl, p in zip(logging_percentiles, pct):
            metrics[f"lifespan/p{int(pl)}"] = float(p)
    return metrics


def step_diagnostics(grads: list, lm: dict, normalize_advantages: bool) -> dict:
    """The per-gradient-step rows behind GRAD_NORM and CRITIC_EXTRA for a single reward
    key: ppo.py:313-315 (`sqrt(sum g^2)` of the gradients BEFORE clipping),
    ppo.py:520-528 (normalised advantages; R^2 = 1 - 2 * critic_loss / (var(target) + 1e-8))."""
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads))
    adv = lm["advantages"]
    a = adv
    if normalize_advantages:
        a = (a - a.mean()) / (a.std(unbiased=False) + 1e-8)
    target = lm["values"] + adv
    r2 = 1.0 - 2.0 * lm["critic"] / (target.var(unbiased=False) + 1e-8)
    return {"grad_norm": gn, "advantages": a, "critic_R^2": r2}
