"""Env x network rollout with reset-on-done (counterpart of
`nnx_ppo/algorithms/rollout.py`: `single_transition` 11-45, `unroll_env` 48-73,
`eval_rollout` 97-148, `tree_where` 270-279)."""
from __future__ import annotations

from typing import Any, Optional

import torch

from .. import ops
from .. import random as rnd
from ..envs.constants import cast_constant, is_constant
from ..networks.types import StatefulModule
from ..tree import tree_all, tree_map
from .types import Transition


def collect_pairs(B: int, on_true: Any, on_false: Any, special=None):
    """The leaf walk of `tree_where`: returns (pairs, skeleton).  `pairs[i]` = the
    contiguous (on_true, on_false) tensors of a leaf that is selected per row; the
    skeleton holds `_Slot(i)` there, the `on_true` leaf where the reference's shared-field
    rule applies (rollout.py:272-275), and — `special(x, y)` returning not-None — whatever
    that callback returns (leaves a fused kernel produces itself)."""
    pairs: list = []

    def collect(x, y):
        if special is not None:
            s = special(x, y)
            if s is not None:
                return s
        if not isinstance(x, torch.Tensor) or x.dim() == 0 or x.shape[0] != B:
            return x
        if not isinstance(y, torch.Tensor):
            return x
        if y.dtype != x.dtype:
            if is_constant(x):
                x = cast_constant(x, y.dtype)  # cached: no launch (e.g. reset's done=False)
            else:
                y = y.to(x.dtype)
        pairs.append((x.contiguous(), y.contiguous()))
        return _Slot(len(pairs) - 1)

    return pairs, tree_map(collect, on_true, on_false)


def fill_slots(skeleton: Any, outs: list, special=None) -> Any:
    def put(v):
        if isinstance(v, _Slot):
            return outs[v.i]
        if special is not None:
            return special(v)
        return v

    return tree_map(put, skeleton)


def tree_where(cond: torch.Tensor, on_true: Any, on_false: Any) -> Any:
    """rollout.py:270-279 — per leaf `where(cond[:, None...], x, y)`; leaves whose
    leading dim is not the batch are taken from `on_true` unchanged (the
    reference's shared-field rule, 272-275).  One byte-exact select launch for all
    leaves (`mi_select_rows_multi`)."""
    pairs, skeleton = collect_pairs(cond.shape[0], on_true, on_false)
    return fill_slots(skeleton, ops.select_rows_multi(cond, pairs))


def stack_steps(steps: list) -> Any:
    """`tree_map(lambda *xs: stack(xs, 0), *steps)` — what the reference's scan does with
    its per-step outputs (rollout.py:61-66) — with every GPU leaf in ONE launch
    (`mi_stack_multi`) instead of one `torch.cat` launch per leaf."""
    groups: list = []

    def collect(*xs):
        x0 = xs[0]
        if (isinstance(x0, torch.Tensor) and x0.is_cuda
                and all(x.shape == x0.shape and x.dtype == x0.dtype for x in xs)):
            groups.append([x if x.is_contiguous() else x.contiguous() for x in xs])
            return _Slot(len(groups) - 1)
        return torch.stack(xs, dim=0)

    skeleton = tree_map(collect, steps[0], *steps[1:])
    return fill_slots(skeleton, ops.stack_multi(groups))


class _Slot:
    def __init__(self, i: int):
        self.i = i


def _as_bool(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == torch.bool:
        return x
    flag = getattr(x, "done_flag", None)  # EpisodeWrapper.step emits both forms at once
    return flag if flag is not None else x != 0


def single_transition(env, networks: StatefulModule, carry, rng_keys_for_env_reset,
                      reset_states=None):
    """rollout.py:11-45.  `reset_states` (optional): `env.reset(rng_keys_for_env_reset)`
    evaluated ahead of time (see `unroll_env`)."""
    network_state, env_state = carry
    out = networks(network_state, env_state.obs)
    next_network_state = out.next_state
    ppo_output = out.output
    # an env that can step AND apply the reset-on-done select in one launch
    # (wrappers/episode_wrapper.py) hands back both states
    reset_env_state = None
    step_and_reset = getattr(env, "step_and_reset", None) if reset_states is not None else None
    if step_and_reset is not None:
        next_env_state, reset_env_state = step_and_reset(env_state, ppo_output.actions,
                                                         reset_states)
    else:
        next_env_state = env.step(env_state, ppo_output.actions)
    done = _as_bool(next_env_state.done)
    trunc = next_env_state.info.get("truncated", None)
    trunc = torch.zeros_like(done) if trunc is None else _as_bool(trunc)
    transition = Transition(
        obs=env_state.obs,
        network_output=ppo_output,
        rewards=next_env_state.reward,
        done=done,
        truncated=trunc,
        next_obs=next_env_state.obs,
        metrics={"env": next_env_state.metrics, "net": out.metrics},
        rollout_extras=out.rollout_extras,
    )
    if reset_env_state is not None:
        next_env_state = reset_env_state
    else:
        if reset_states is None:
            reset_states = env.reset(rng_keys_for_env_reset)
        next_env_state = tree_where(done, reset_states, next_env_state)
    reset_network_states = networks.reset_state(next_network_state)
    next_network_state = tree_where(done, reset_network_states, next_network_state)
    return (next_network_state, next_env_state), transition


def unroll_env(env, env_state, networks: StatefulModule, network_state, unroll_length: int,
               rng_key_for_env_reset: torch.Tensor):
    """rollout.py:48-73 — returns (final_network_state, final_env_state, Transition
    with time-major `[T, N, ...]` leaves)."""
    # a network / env pair that can run the whole scan in one launch (networks/policy.py:
    # MLPActorCritic.unroll_fused over EpisodeWrapper(MockEnv)) — every leaf bit-identical to
    # the steps below
    fused = getattr(networks, "unroll_fused", None)
    if fused is not None:
        res = fused(env, env_state, network_state, unroll_length, rng_key_for_env_reset)
        if res is not None:
            return res
    batch_size = env_state.done.shape[0]
    keys = rnd.split(rng_key_for_env_reset, (unroll_length, batch_size))
    # The reset state of step t is a function of keys[t] alone (rollout.py:57-59
    # evaluates it inside the scan, every step, for every env): evaluate all T of
    # them in ONE batched call — same values, T times fewer launches on the
    # per-step critical path.
    TN = unroll_length * batch_size
    resets = env.reset(keys.reshape(TN))

    def at_step(t):
        return tree_map(
            lambda x: x.view(unroll_length, batch_size, *x.shape[1:])[t]
            if isinstance(x, torch.Tensor) and x.dim() >= 1 and x.shape[0] == TN else x,
            resets)

    carry = (network_state, env_state)
    steps = []
    for t in range(unroll_length):
        carry, tr = single_transition(env, networks, carry, keys[t], reset_states=at_step(t))
        steps.append(tr)
    rollout = stack_steps(steps)
    shapes_match = tree_map(lambda v, r: v.shape == r.shape,
                            rollout.network_output.value_estimates, rollout.rewards)
    assert tree_all(shapes_match), "value_estimates leaves must match rewards leaves"
    return carry[0], carry[1], rollout


def _add_reward_metrics(out: dict, name: str, reward: Any,
                        percentile_levels: Optional[tuple]) -> None:
    """rollout.py:76-94."""
    if isinstance(reward, dict):
        for k, v in reward.items():
            _add_reward_metrics(out, f"{name}/{k}", v, percentile_levels)
    elif percentile_levels is not None:
        from .metrics import percentiles

        pct = percentiles(reward, tuple(percentile_levels))
        for pl, p in zip(percentile_levels, pct):
            out[f"{name}/p{int(pl)}"] = p
    else:
        out[f"{name}/mean"] = reward.mean()
        out[f"{name}/std"] = reward.std(unbiased=False)


def eval_rollout(env, networks: StatefulModule, n_envs: int, max_episode_length: int,
                 key: torch.Tensor, logging_percentiles: Optional[tuple] = None) -> dict:
    """rollout.py:97-148 — sticky-done evaluation: reward accumulates until the
    first done of each env; lifespan counts the steps before it."""
    env_states = env.reset(rnd.split(key, n_envs))
    net_states = networks.initialize_state(n_envs)
    cuml_reward = tree_map(torch.zeros_like, env_states.reward)
    dev = env_states.done.device
    lifespan = torch.zeros(n_envs, dtype=torch.float32, device=dev)
    env_state = env_states
    for _ in range(max_episode_length):
        out = networks(net_states, env_state.obs)
        net_states = out.next_state
        nxt = env.step(env_state, out.output.actions)
        prev_done = _as_bool(env_state.done)
        sticky = torch.logical_or(_as_bool(nxt.done), prev_done)
        nxt = nxt.replace(done=sticky.to(torch.float32))
        reward_this_step = tree_map(lambda r: torch.where(prev_done, torch.zeros_like(r), r),
                                    nxt.reward)
        cuml_reward = tree_map(torch.add, cuml_reward, reward_this_step)
        lifespan = lifespan + torch.where(sticky, 0.0, 1.0)
        env_state = nxt
    metrics = dict(lifespan_mean=lifespan.mean(), lifespan_std=lifespan.std(unbiased=False))
    _add_reward_metrics(metrics, "episode_reward", cuml_reward, logging_percentiles)
    if logging_percentiles is not None:
        from .metrics import percentiles

        for pl, p in zip(logging_percentiles,
                         percentiles(lifespan, tuple(logging_percentiles))):
            metrics[f"lifespan/p{int(pl)}"] = p
    return metrics
