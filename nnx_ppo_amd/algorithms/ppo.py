"""PPO driver, one-iteration step, GAE and loss (counterpart of
`nnx_ppo/algorithms/ppo.py`): `train_ppo` 41-251, `ppo_step` 254-348, `gae`
351-394, `ppo_loss` 397-531, `new_training_state` 534-572, `_should_run` 34-38 —
same names, signatures, defaults and semantics, with the arithmetic done by the
HIP kernels of libmippo:

  * advantages / targets are recomputed inside every minibatch loss from the
    CURRENT parameters' values (ppo.py:447-458);
  * minibatches are over envs only and replay starts from the PRE-rollout
    carry (ppo.py:297-300);
  * the entropy regulariser is a one-sample estimate with fresh noise per loss
    evaluation (sampling_layers.py:137-147);
  * normaliser statistics are frozen during the iteration and updated last
    (ppo.py:336).
"""
from __future__ import annotations

import contextlib
import dataclasses
import os
import time
from collections.abc import Callable
from typing import Any, Optional

import torch

from .. import config as mi_config
from .. import ops, parallel
from .. import random as rnd
from ..networks.adapter import _Fork, _can_fork
from ..networks.types import PPONetworkOutput, StatefulModule, bump_param_epoch
from ..optim import Optimizer
from ..tree import tree_leaves, tree_map
from . import rollout
from .config import (BackendConfig, EvalConfig, PPOConfig, TrainConfig,  # noqa: F401
                     TrainResult, VideoConfig, VideoData)
from .loop import IterationRunner, run_training_loop, should_run
from .metrics import compute_metrics, log_weight_stats
from .types import LoggingLevel, TrainingState, Transition


# GAE + loss as ONE launch (csrc/gae_loss.hip; 12.6 us against 15.1 us for the two launches
# at [30, 1024], tools/microbench_gae_loss.py).  MIPPO_FUSED_GAE_LOSS=0: two launches (A/B
# timing; also what a sharded run uses — the advantage statistics are exchanged in between).
FUSED_GAE_LOSS = os.environ.get("MIPPO_FUSED_GAE_LOSS", "1") != "0"
# MIPPO_SHARDED_GAE_IN_BWD=0: a sharded run keeps GAE / statistics exchange / loss as launches
# of their own in front of the backward (A/B, and the reference form of the equivalence tests)
SHARDED_GAE_IN_BWD = os.environ.get("MIPPO_SHARDED_GAE_IN_BWD", "1") != "0"


def default_config() -> TrainConfig:
    return TrainConfig()


def _should_run(steps: int, last_step: int, every_steps: int) -> bool:
    """ppo.py:34-38."""
    return should_run(steps, last_step, every_steps)


def train_ppo(
    env,
    networks: StatefulModule,
    config: Optional[TrainConfig] = None,
    *,
    total_steps: Optional[int] = None,
    seed: Optional[int] = None,
    log_fn: Optional[Callable[[dict, int], None]] = None,
    video_fn: Optional[Callable[[VideoData], None]] = None,
    checkpoint_fn: Optional[Callable[[TrainingState, int], None]] = None,
    eval_env=None,
    initial_state: Optional[TrainingState] = None,
    compute_dtype: Optional[str] = None,
    hip_graph: Optional[bool] = None,
    overlap_logging: Optional[bool] = None,
) -> TrainResult:
    """ppo.py:41-251.  `networks` is trained in place.

    The three trailing keywords (and `TrainConfig.backend`, which they override) are this
    build's counterpart of what `nnx.jit` decides for the reference:
      compute_dtype   "f32" | "bf16": MFMA path of the Dense layers (BASELINE configs[1]
                      is quoted in bf16); default: the process-wide `nnx_ppo_amd.config`.
      hip_graph       True (default): iteration 1 runs eagerly, iteration 2 is recorded
                      into a HIP graph, later ones are one `hipGraphLaunch` each — the
                      role of `nnx.jit(ppo_step)` (ppo.py:105).  An env or module that
                      cannot be captured raises `GraphCaptureError`; pass False for it.
      overlap_logging True (default): iteration i+1 is enqueued before the host waits
                      for iteration i's metrics whenever no eval / video / checkpoint is
                      due in between (`algorithms/loop.py`).
    `log_fn` receives 0-d CPU tensors (one device-to-host copy per iteration — the host
    sync of ppo.py:209)."""
    if config is None:
        config = default_config()
    if total_steps is not None:
        config = dataclasses.replace(config,
                                     ppo=dataclasses.replace(config.ppo, total_steps=total_steps))
    if seed is not None:
        config = dataclasses.replace(config, seed=seed)
    if eval_env is None:
        eval_env = env
    backend = config.backend
    if compute_dtype is None:
        compute_dtype = backend.compute_dtype
    if hip_graph is None:
        hip_graph = backend.hip_graph
    if overlap_logging is None:
        overlap_logging = backend.overlap_logging
    dtype_ctx = (mi_config.use_compute_dtype(compute_dtype) if compute_dtype is not None
                 else contextlib.nullcontext())
    with dtype_ctx:
        return _train_ppo(env, networks, config, log_fn, video_fn, checkpoint_fn, eval_env,
                          initial_state, bool(hip_graph), bool(overlap_logging))


def _train_ppo(env, networks, config, log_fn, video_fn, checkpoint_fn, eval_env, initial_state,
               hip_graph: bool, overlap: bool) -> TrainResult:
    if initial_state is None:
        training_state = new_training_state(
            env, networks, config.ppo.n_envs, config.seed, config.ppo.learning_rate,
            config.ppo.gradient_clipping, config.ppo.weight_decay)
    else:
        training_state = initial_state
    device = training_state.steps_taken.device

    eval_history: list[dict] = []
    last_eval_step = -config.eval.every_steps
    last_video_step = -config.video.every_steps
    last_checkpoint_step = -config.checkpoint_every_steps
    metrics: dict = {}
    measure_throughput = LoggingLevel.THROUGHPUT in config.ppo.logging_level

    def run_eval(steps: int) -> dict:
        networks.eval()
        t0 = time.perf_counter() if measure_throughput else None
        eval_metrics = rollout.eval_rollout(
            eval_env, networks, config.eval.n_envs, config.eval.max_episode_length,
            rnd.key(config.seed, device), config.eval.logging_percentiles)
        if measure_throughput:
            if device.type == "cuda":
                torch.cuda.synchronize(device)
            elapsed = time.perf_counter() - t0
            eval_metrics = dict(eval_metrics)
            eval_metrics["throughput/eval_sps"] = (
                config.eval.n_envs * config.eval.max_episode_length / elapsed)
        networks.train()
        return dict(eval_metrics)

    def run_video(steps: int, iteration: int) -> dict:
        # Rendering needs a MuJoCo renderer (rollout.py:150-267); out of scope.
        return {}

    steps = int(training_state.steps_taken)
    if config.eval.enabled:
        eval_metrics = run_eval(steps)
        metrics.update(eval_metrics)
        eval_history.append({"step": steps, **eval_metrics})
        last_eval_step = steps
    if config.video.enabled:
        metrics.update(run_video(steps, 0))
        last_video_step = steps
    if checkpoint_fn is not None and _should_run(steps, last_checkpoint_step,
                                                 config.checkpoint_every_steps):
        checkpoint_fn(training_state, steps)
        last_checkpoint_step = steps
    if log_fn is not None and metrics:
        log_fn(metrics, steps)

    c = config.ppo
    step_fn = lambda ts: ppo_step(
        env, ts, c.n_envs, c.rollout_length, c.gae_lambda, c.discounting_factor, c.clip_range,
        c.normalize_advantages, c.combine_advantages, c.n_epochs, c.n_minibatches,
        c.critic_loss_weight, c.logging_level, c.logging_percentiles)
    runner = IterationRunner(step_fn, training_state, hip_graph=hip_graph,
                             networks=[networks])
    # `steps_taken` advances by the same constant every iteration (ppo.py:338), so the loop
    # counts it on the host and checks it against the device counter once, at the end
    loop_metrics, steps, n_iterations = run_training_loop(
        runner, total_steps=c.total_steps, steps=steps,
        steps_per_iteration=c.rollout_length * c.n_envs * parallel.world_size(),
        local_steps_per_iteration=c.rollout_length * c.n_envs,
        measure_throughput=measure_throughput,
        eval_every=config.eval.every_steps, video_every=config.video.every_steps,
        checkpoint_every=config.checkpoint_every_steps,
        last_eval_step=last_eval_step, last_video_step=last_video_step,
        last_checkpoint_step=last_checkpoint_step,
        run_eval=run_eval if config.eval.enabled else None,
        run_video=run_video if config.video.enabled else None,
        checkpoint_fn=checkpoint_fn, log_fn=log_fn, eval_history=eval_history,
        overlap=overlap)
    if n_iterations:
        metrics = loop_metrics
    training_state = runner.state
    device_steps = int(training_state.steps_taken)  # the one host read of the counter
    if device_steps != steps:  # the host counted the iterations; the device counted the steps
        raise RuntimeError(f"the device's step counter ({device_steps}) differs from the "
                           f"host's count ({steps}): an iteration was lost or ran twice")
    return TrainResult(
        training_state=training_state,
        final_metrics=metrics,
        eval_history=eval_history,
        total_steps=device_steps,
        total_iterations=n_iterations,
    )


def minibatch_indices(new_key: torch.Tensor, n_envs: int, n_epochs: int,
                      n_minibatches: int) -> torch.Tensor:
    """ppo.py:284-294 — per epoch a permutation of the envs (time axis kept
    whole), reshaped `[n_minibatches, n_envs // n_minibatches]`; int64
    `[n_epochs * n_minibatches, minibatch_size]`."""
    minibatch_size = n_envs // n_minibatches
    perms = rnd.permutations(new_key, n_epochs, n_envs)  # [n_epochs, n_envs]
    used = n_minibatches * minibatch_size
    return perms[:, :used].reshape(n_epochs * n_minibatches, minibatch_size)


def _advance_noise(networks: StatefulModule) -> None:
    for m in networks.modules():
        adv = getattr(m, "advance_rng", None)
        if adv is not None:
            adv()


def ppo_step(
    env,
    training_state: TrainingState,
    n_envs: int,
    rollout_length: int,
    gae_lambda: float,
    discounting_factor: float,
    clip_range: float,
    normalize_advantages: bool,
    combine_advantages: bool,
    n_epochs: int,
    n_minibatches: int,
    critic_loss_weight: float = 1.0,
    logging_level: LoggingLevel = LoggingLevel.LOSSES,
    logging_percentiles: Optional[tuple] = None,
    *,
    minibatch_inds: Optional[torch.Tensor] = None,
) -> tuple[TrainingState, dict]:
    """ppo.py:254-348.  `minibatch_inds` (optional, int64
    `[n_epochs*n_minibatches, mb]`) injects the minibatch indices instead of
    drawing them from the key (parity tests)."""
    networks: StatefulModule = training_state.networks
    optimizer: Optimizer = training_state.optimizer
    # derived parameter copies (bf16 shadows) are refreshed at first use in every
    # iteration, so a captured graph never replays with stale ones
    bump_param_epoch()

    keys = rnd.split(training_state.rng_key)
    reset_key, new_key = keys[0], keys[1]
    total_iterations = n_epochs * n_minibatches
    if minibatch_inds is None:
        # every epoch's permutation in one launch (they depend on the key only).  Not
        # forked onto the second stream: a second branch in the captured graph costs
        # more than the kernel it would hide (measured: 31.2 M vs 32.5 M env-steps/s).
        all_indices = minibatch_indices(new_key, n_envs, n_epochs, n_minibatches)
    else:
        all_indices = minibatch_inds
    assert all_indices.shape[0] == total_iterations

    next_net_state, next_env_state, rollout_data = rollout.unroll_env(
        env, training_state.env_states, networks, training_state.network_states,
        rollout_length, reset_key)

    # only the Transition fields the loss reads are gathered (ppo.py:297 gathers
    # every leaf; `metrics`, `actions`, `value_estimates` are dead there)
    loss_view = Transition(
        obs=rollout_data.obs,
        network_output=PPONetworkOutput(
            actions=None, loglikelihoods=rollout_data.network_output.loglikelihoods,
            value_estimates=None),
        rewards=rollout_data.rewards, done=rollout_data.done,
        truncated=rollout_data.truncated,
        next_obs=tree_map(lambda x: x[-1:], rollout_data.next_obs),
        metrics={}, rollout_extras=rollout_data.rollout_extras)

    device = rollout_data.done.device
    # every entry is written by the loss launch of its gradient step: no zero-fill launch
    loss_rows = torch.empty(total_iterations, 4, dtype=torch.float32, device=device)
    grad_norms = None
    if LoggingLevel.GRAD_NORM in logging_level:
        grad_norms = torch.empty(total_iterations, dtype=torch.float32, device=device)

    critic_extra: dict = {}
    pending_losses: list = []
    # minibatch gather x[:, inds] (ppo.py:297-300).  The indices of every gradient step are
    # known before the first one, and the gathered leaves (rollout data, pre-rollout
    # carry) do not change during the update: ONE launch gathers all n_epochs *
    # n_minibatches minibatches, each a contiguous time-major block, instead of one launch
    # per gradient step.  (Memory: n_epochs x the loss's share of the rollout; above
    # 1 GiB the gather goes back to one launch per step.)
    mb_leaves = tree_leaves(loss_view)
    st_leaves = tree_leaves(training_state.network_states)
    gather_src = mb_leaves + [x.unsqueeze(0) for x in st_leaves]
    mb_size = all_indices.shape[1]
    per_step_bytes = sum(x[:, :1].numel() * x.element_size() for x in gather_src) * mb_size
    gather_all = per_step_bytes * total_iterations <= (1 << 30)
    if gather_all:
        all_gathered = ops.gather_cols_multi(gather_src, all_indices.reshape(-1).contiguous(),
                                             groups=total_iterations)
    for i in range(total_iterations):
        if gather_all:
            gathered = [g[i] for g in all_gathered]
        else:
            gathered = ops.gather_cols_multi(gather_src, all_indices[i].contiguous())
        it = iter(gathered)
        minibatch = tree_map(lambda x: next(it), loss_view)
        net_state_subset = tree_map(lambda x: next(it).squeeze(0),
                                    training_state.network_states)
        optimizer.begin(defer_dw=True)
        _, lm = ppo_loss(networks, net_state_subset, minibatch, clip_range,
                         normalize_advantages, combine_advantages, discounting_factor,
                         gae_lambda, critic_loss_weight, logging_level, loss_out=loss_rows[i],
                         want_total=False, defer_loss=pending_losses)
        for k, v in lm.items():
            # per-step diagnostics that are not columns of `loss_rows`: CRITIC_EXTRA, and
            # the per-key trees of a PyTree reward / value / log-likelihood setup
            if k in ("losses/advantages", "losses/critic_R^2") or not isinstance(v, torch.Tensor):
                critic_extra.setdefault(k, []).append(v)
        # GRAD_NORM (ppo.py:313-315): the norm of the gradient the optimiser applies,
        # written straight into its row (sharded: taken after the all-reduce)
        optimizer.update(norm_out=None if grad_norms is None else grad_norms[i:i + 1])

    ops.policy_loss_finalize(pending_losses)  # the loss rows left as per-tile partials
    scale = 1.0
    if parallel.is_distributed():
        parallel.allreduce_sum_(loss_rows)
        scale = 1.0 / parallel.world_size()

    # metrics.py:72-100 for the loss rows: mean / population std over the gradient steps of
    # all four columns in ONE launch (percentiles, when asked for, go through `_log_metric`)
    names = []
    if LoggingLevel.LOSSES in logging_level:
        names += [(0, "losses/actor"), (1, "losses/critic"), (2, "losses/regularization")]
    if LoggingLevel.ACTOR_EXTRA in logging_level:
        names += [(3, "losses/clipping_fraction")]
    loss_metrics: dict = {}
    ready: dict = {}
    if names and not logging_percentiles:
        ms = ops.col_mean_std(loss_rows, scale)
        for col, name in names:
            ready[f"{name}/mean"] = ms[0, col]
            ready[f"{name}/std"] = ms[1, col]
    else:
        rows = loss_rows if scale == 1.0 else loss_rows * scale
        for col, name in names:
            loss_metrics[name] = rows[:, col]
    if grad_norms is not None:
        loss_metrics["grad_norm"] = grad_norms
    for k, v in critic_extra.items():  # stacked over the gradient steps, as the scan does
        loss_metrics[k] = tree_map(lambda *xs: torch.stack(xs, dim=0), v[0], *v[1:])
        # a per-key tree (PyTree rewards / values / log-likelihoods) replaces the summed
        # column of the same name
        ready.pop(f"{k}/mean", None)
        ready.pop(f"{k}/std", None)

    total_steps = training_state.steps_taken + rollout_length * n_envs * parallel.world_size()
    metrics = dict(ready)
    metrics.update(compute_metrics(loss_metrics, rollout_data, logging_level,
                                   logging_percentiles))
    metrics["total_steps"] = total_steps
    if LoggingLevel.WEIGHTS in logging_level:
        # the parameters themselves, not the arena (its alignment padding is zeros)
        log_weight_stats(metrics, torch.cat([p.data.reshape(-1) for p in networks.parameters()]),
                         logging_percentiles)
    networks.update_statistics(rollout_data.rollout_extras)
    _advance_noise(networks)

    training_state = training_state.replace(
        network_states=next_net_state,
        env_states=next_env_state,
        rng_key=new_key,
        steps_taken=total_steps,
    )
    return training_state, metrics


def gae(rewards: torch.Tensor, values_excl_last: torch.Tensor, last_value: torch.Tensor,
        done: torch.Tensor, truncation: torch.Tensor, lambda_: float,
        gamma: float) -> torch.Tensor:
    """ppo.py:351-394 (same argument order) — one launch of `mi_gae_f32`."""
    return ops.gae(rewards.contiguous(), values_excl_last.contiguous(),
                   last_value.contiguous(), done.contiguous(), truncation.contiguous(),
                   gamma, lambda_)


def ppo_loss(
    networks: StatefulModule,
    network_state: Any,
    rollout_data: Transition,
    clip_range: float,
    normalize_advantages: bool,
    combine_advantages: bool,
    discounting_factor: float,
    gae_lambda: float,
    critic_loss_weight: float,
    logging_level: LoggingLevel,
    *,
    loss_out: Optional[torch.Tensor] = None,
    backward: bool = True,
    want_total: bool = True,
    defer_loss: Optional[list] = None,
) -> tuple[Optional[torch.Tensor], dict]:
    """ppo.py:397-531.  Evaluates the loss on one minibatch (`[T, mb, ...]` leaves)
    and — where the reference returns gradients from `nnx.grad` — ACCUMULATES the
    parameter gradients into `Parameter.grad`.  Returns (total_loss, loss_metrics);
    `loss_out` (float32 [4]) receives (actor, critic, regularization,
    clipping_fraction) without a host sync.  `defer_loss` (a list, with `loss_out` and
    `want_total=False`): a path that can leave the four scalars as partial sums does so and
    appends to the list; they are filled by `ops.policy_loss_finalize(defer_loss)` — `ppo_step`
    does that once per iteration instead of 16 times at the tail of a launch."""
    done = rollout_data.done
    truncated = rollout_data.truncated
    T, B = done.shape

    # bootstrap value at T (ppo.py:433-437): forward on the last next_obs; only the
    # value estimate is used, so only the value port is evaluated
    last_obs = tree_map(lambda x: x[-1], rollout_data.next_obs)
    stateless = not any(isinstance(t, torch.Tensor) for t in tree_leaves(network_state))
    fork = None
    fused = None
    if hasattr(networks, "replay_with_bootstrap"):
        # the bootstrap rows ride along in the replay launch (networks/policy.py for the
        # fused MLP class, containers.Sequential for a stateless Dense value port beside any
        # action port)
        fused = networks.replay_with_bootstrap(network_state, rollout_data.obs, done,
                                               rollout_data.rollout_extras, last_obs)
    if fused is not None:
        ctx, out, reg_seq, final_state, last_values = fused
    elif stateless and _can_fork(last_obs, any(getattr(m, "_wide", False)
                                               for m in networks.modules())):
        # no carry: the bootstrap forward does not depend on the replay's final
        # state, so it runs beside the replay on the second stream (it is a 1/T-size
        # launch that would otherwise sit alone on the critical path)
        fork = _Fork(last_obs)
        with fork:
            last_values = networks.forward_value(network_state, last_obs)
    if fused is None:
        # replay scan (ppo.py:411-431), layer by layer over the whole sequence
        ctx, out, reg_seq, final_state = networks.replay(
            network_state, rollout_data.obs, done, rollout_data.rollout_extras,
            need_input_grad=False)
        if fork is not None:
            fork.join(last_values)
        else:
            last_values = networks.forward_value(final_state, last_obs)

    rewards = rollout_data.rewards
    values = out.value_estimates
    ll_new = out.loglikelihoods
    ll_old = rollout_data.network_output.loglikelihoods
    single = (isinstance(rewards, torch.Tensor) and isinstance(values, torch.Tensor)
              and isinstance(ll_new, torch.Tensor))
    if not single:
        g_ll, g_v, loss_out, detail = _pytree_loss_terms(
            rewards, values, last_values, ll_new, ll_old, reg_seq, done, truncated,
            clip_range, normalize_advantages, combine_advantages, discounting_factor,
            gae_lambda, critic_loss_weight, loss_out, logging_level)
        if backward:
            networks.replay_backward(
                ctx, PPONetworkOutput(actions=None, loglikelihoods=g_ll, value_estimates=g_v),
                1.0 / float(T * B))
    else:
        del combine_advantages  # single reward key: nothing to combine
        values = values.contiguous()
        reg_flat = None if reg_seq is None else reg_seq.reshape(-1)
        fused_loss = (FUSED_GAE_LOSS and not parallel.is_distributed()
                      and ops.gae_ppo_loss_supported(T, B))
        # sharded over the one-shot transport the statistics cross the ranks INSIDE the
        # backward launch (mi_policy_ws_bwd_gae_bf16 with the communicator): a sharded rank
        # launches what a single GPU launches.  Over RCCL (host collectives) the three
        # launches with the all-reduce between them remain.
        in_bwd_ok = fused_loss or (FUSED_GAE_LOSS and parallel.peer_comm() is not None
                                   and SHARDED_GAE_IN_BWD)
        gae_bwd = (in_bwd_ok and backward and LoggingLevel.CRITIC_EXTRA not in logging_level
                   and fused is not None and hasattr(networks, "replay_backward_gae")
                   and networks.gae_backward_supported(ctx, T, B))
        if gae_bwd:
            # no launch at all between the replay forward and the backward: the scan, the
            # statistics and the loss gradients are evaluated by the backward's workgroups
            loss_out = networks.replay_backward_gae(
                ctx, 1.0 / float(T * B), rewards.contiguous(), values,
                last_values.contiguous(), done.contiguous(), truncated.contiguous(),
                ll_new.contiguous(), ll_old.contiguous(),
                None if reg_seq is None else reg_seq.contiguous(), discounting_factor,
                gae_lambda, normalize_advantages, clip_range, critic_loss_weight,
                loss_out=loss_out,
                defer=defer_loss if (loss_out is not None and not want_total) else None)
            backward = False
        elif fused_loss:
            # GAE, advantage statistics, loss terms and gradients in ONE launch (the
            # advantages never leave registers); a sharded run exchanges the statistics
            # between the two phases and keeps the two launches
            g_ll, g_v, loss_out, adv = ops.gae_ppo_loss(
                rewards.contiguous(), values, last_values.contiguous(), done.contiguous(),
                truncated.contiguous(), ll_new.contiguous(), ll_old.contiguous(),
                None if reg_seq is None else reg_seq.contiguous(), discounting_factor,
                gae_lambda, normalize_advantages, clip_range, critic_loss_weight,
                loss_out=loss_out, want_adv=LoggingLevel.CRITIC_EXTRA in logging_level,
                defer=defer_loss if (loss_out is not None and not want_total
                                     and LoggingLevel.CRITIC_EXTRA not in logging_level)
                else None)
        else:
            stats = None
            if normalize_advantages:
                # the statistics of ppo.py:477-480 come out of the GAE launch itself
                adv, stats = ops.gae(rewards.contiguous(), values, last_values.contiguous(),
                                     done.contiguous(), truncated.contiguous(),
                                     discounting_factor, gae_lambda, with_stats=True)
                if parallel.is_distributed():
                    parallel.allreduce_sum_(stats)
            else:
                adv = ops.gae(rewards.contiguous(), values, last_values.contiguous(),
                              done.contiguous(), truncated.contiguous(), discounting_factor,
                              gae_lambda)
            g_ll, g_v, loss_out = ops.ppo_loss(
                ll_new.reshape(-1), ll_old.reshape(-1), adv.reshape(-1), values.reshape(-1),
                reg_flat, stats, clip_range, critic_loss_weight, loss_out=loss_out)
        if backward:
            g_out = PPONetworkOutput(actions=None, loglikelihoods=g_ll.view(T, B),
                                     value_estimates=g_v.view(T, B))
            networks.replay_backward(ctx, g_out, 1.0 / float(T * B))
        if LoggingLevel.CRITIC_EXTRA in logging_level:
            # ppo.py:520-528 (diagnostics only: plain torch reductions on the side)
            a = adv
            if normalize_advantages:
                a = (a - a.mean()) / (a.std(unbiased=False) + 1e-8)
            target = values + adv
            extra = {"losses/advantages": a,
                     "losses/critic_R^2": 1.0 - 2.0 * loss_out[1] /
                     (target.var(unbiased=False) + 1e-8)}

    loss_metrics: dict = {}
    if LoggingLevel.CRITIC_EXTRA in logging_level and single:
        loss_metrics.update(extra)
    if LoggingLevel.LOSSES in logging_level:
        loss_metrics["losses/actor"] = loss_out[0]
        loss_metrics["losses/critic"] = loss_out[1]
        loss_metrics["losses/regularization"] = loss_out[2]
    if LoggingLevel.ACTOR_EXTRA in logging_level:
        loss_metrics["losses/clipping_fraction"] = loss_out[3]
    if not single:
        loss_metrics.update(detail)  # per-key trees replace the summed scalars
    total = None
    if want_total:  # three tiny launches: the training loop reads `loss_out` instead
        total = loss_out[0] + critic_loss_weight * loss_out[1] + loss_out[2]
    return total, loss_metrics


def _pytree_loss_terms(rewards, values, last_values, ll_new, ll_old, reg_seq, done, truncated,
                       clip_range, normalize_advantages, combine_advantages, gamma, lambda_,
                       critic_loss_weight, loss_out, logging_level=LoggingLevel.NONE):
    """ppo.py:440-510 for PyTree rewards / value heads / log-likelihoods.

    One GAE per reward key (`done` / `truncated` are shared, ppo.py:440-445); the critic
    term of every key against its own target; the actor term of every log-likelihood
    leaf against its advantage — the key's own, or, with `combine_advantages`, the sum
    over keys (broadcast to every leaf of a structured log-likelihood tree,
    ppo.py:462-474); each advantage tree leaf is normalised on its own (477-480).
    One loss launch per leaf; the scalars are summed (`jax.tree.reduce`, 505-507).
    Returns (g_ll tree, g_v tree, loss_out[4] = summed actor, critic, reg, mean clip
    fraction, detail = the per-key metric trees of ppo.py:509-528)."""
    T, B = done.shape
    d_, t_ = done.contiguous(), truncated.contiguous()
    adv = tree_map(
        lambda r, v, lv: ops.gae(r.contiguous(), v.contiguous(), lv.contiguous(), d_, t_, gamma,
                                 lambda_),
        rewards, values, last_values)
    dev = done.device
    if loss_out is None:
        loss_out = torch.empty(4, dtype=torch.float32, device=dev)
    parts = []
    detail: dict = {}
    # critic terms, per reward key, on the raw advantages

    def critic(v, a):
        _, gv, lo = ops.ppo_loss(None, None, a.reshape(-1), v.contiguous().reshape(-1), None,
                                 None, clip_range, critic_loss_weight)
        parts.append(lo)
        return gv.view(T, B), lo

    cr = tree_map(critic, values, adv)
    g_v = _tree_pick(cr, values, 0)
    critic_rows = _tree_pick(cr, values, 1)
    # actor advantages
    if combine_advantages:
        leaves = tree_leaves(adv)
        summed = leaves[0]
        for a in leaves[1:]:
            summed = summed + a
        actor_adv = summed if isinstance(ll_new, torch.Tensor) else \
            tree_map(lambda _: summed, ll_new)
    else:
        actor_adv = adv
        same = (isinstance(ll_new, dict) and isinstance(adv, dict)
                and set(ll_new) == set(adv)) or \
            (isinstance(ll_new, torch.Tensor) and isinstance(adv, torch.Tensor))
        if not same:
            raise ValueError(
                "ppo_loss: the log-likelihood tree and the per-reward-key advantage tree do "
                "not match (ppo.py:494-499 maps over both); give one policy term per reward "
                "key or set combine_advantages=True")
    reg_left = [None if reg_seq is None else reg_seq.reshape(-1)]

    def actor(ln, lo_, a):
        stats = None
        if normalize_advantages:
            stats = ops.adv_stats(a.contiguous())
            if parallel.is_distributed():
                parallel.allreduce_sum_(stats)
        reg, reg_left[0] = reg_left[0], None  # the (shared) regulariser is counted once
        gl, _, lo = ops.ppo_loss(ln.contiguous().reshape(-1), lo_.contiguous().reshape(-1),
                                 a.contiguous().reshape(-1), None, reg, stats, clip_range,
                                 critic_loss_weight)
        parts.append(lo)
        return gl.view(T, B), lo

    ac = tree_map(actor, ll_new, ll_old, actor_adv)
    g_ll = _tree_pick(ac, ll_new, 0)
    actor_rows = _tree_pick(ac, ll_new, 1)
    n_actor = len(tree_leaves(ll_new))
    total = parts[0]
    for p_ in parts[1:]:
        total = total + p_
    loss_out.copy_(total)
    if n_actor > 1:
        loss_out[3:4].div_(float(n_actor))  # clipping fraction: mean over the policy terms
    # the per-key trees the reference logs (ppo.py:509-528)
    if LoggingLevel.LOSSES in logging_level:
        detail["losses/actor"] = tree_map(lambda r: r[0], actor_rows)
        detail["losses/critic"] = tree_map(lambda r: r[1], critic_rows)
    if LoggingLevel.ACTOR_EXTRA in logging_level:
        detail["losses/clipping_fraction"] = tree_map(lambda r: r[3], actor_rows)
    if LoggingLevel.CRITIC_EXTRA in logging_level:
        def norm(a):
            return (a - a.mean()) / (a.std(unbiased=False) + 1e-8) if normalize_advantages else a

        detail["losses/advantages"] = tree_map(norm, actor_adv)
        detail["losses/critic_R^2"] = tree_map(
            lambda r, v, a: 1.0 - 2.0 * r[1] / ((v + a).var(unbiased=False) + 1e-8),
            critic_rows, values, adv)
    return g_ll, g_v, loss_out, detail


def _tree_pick(pairs_tree, like, i: int):
    """`pairs_tree` has the structure of `like` with a tuple at every leaf: pick item i."""
    it = iter(tree_leaves(pairs_tree, is_leaf=lambda x: isinstance(x, tuple)))
    return tree_map(lambda _: next(it)[i], like)


def new_training_state(
    env,
    networks: StatefulModule,
    n_envs: int,
    seed: int,
    learning_rate: float = 1e-4,
    gradient_clipping: Optional[float] = None,
    weight_decay: Optional[float] = None,
    *,
    device=None,
) -> TrainingState:
    """ppo.py:534-572.  Moves `networks` to the device and re-homes its
    parameters into the optimiser's flat arenas.  In a sharded run every rank
    passes the same `seed`: parameters come from the (already constructed)
    network, while env / permutation / noise keys are folded with the rank."""
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError(
                "nnx_ppo_amd needs a GPU: the PPO path runs on HIP kernels and has no CPU "
                "fallback")
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    key = rnd.key(seed, device)
    if parallel.is_distributed():
        key = rnd.fold_in(key, 1 + parallel.rank())
        for m in networks.modules():
            if hasattr(m, "seed") and hasattr(m, "advance_rng"):
                m.seed = (m.seed + 0x9E3779B97F4A7C15 * (1 + parallel.rank())) & (2**63 - 1)
    ks = rnd.split(key)
    key, training_key = ks[0], ks[1]
    env_init_keys = rnd.split(key, n_envs)
    # cloned: envs may hand out shared read-only constants (envs/constants.py) and
    # these leaves become the static, written-in-place buffers of a captured graph
    env_states = tree_map(lambda x: x.clone() if isinstance(x, torch.Tensor) else x,
                          env.reset(env_init_keys))
    networks.to(device)
    network_states = networks.initialize_state(n_envs)
    optimizer = Optimizer(networks, learning_rate, gradient_clipping, weight_decay,
                          device=device)
    return TrainingState(networks, network_states, env_states, optimizer, training_key,
                         torch.zeros((), dtype=torch.int64, device=device))
