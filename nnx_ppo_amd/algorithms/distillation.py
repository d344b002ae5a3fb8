"""Policy distillation (counterpart of `nnx_ppo/algorithms/distillation.py`): the student
is rolled out beside a frozen teacher and trained to maximise
`log p_student(mu_teacher | obs)` — `distillation_single_transition` 66-118,
`distillation_unroll_env` 121-157, `distillation_loss` 160-230, `distillation_step`
233-364, `new_distillation_state` 367-417, `train_distillation` 420-603.

Same skeleton as `ppo_step`, and the same kernels: the rollout engine, the minibatch
gather, the sequence-level `replay` / `replay_backward` protocol and the optimiser; the
loss has no launch of its own — its gradient with respect to every log-likelihood is the
constant -1/(T B) and with respect to every regulariser element +1/(T B)."""
from __future__ import annotations

import contextlib
import dataclasses
from collections.abc import Callable
from typing import Any, Optional

import torch

from .. import config as mi_config
from .. import ops
from .. import random as rnd
from ..networks.types import PPONetworkOutput, StatefulModule, bump_param_epoch
from ..optim import Optimizer
from ..tree import tree_leaves, tree_map
from . import rollout
from .config import (DistillationTrainConfig, DistillationTrainResult, VideoData)
from .loop import IterationRunner, run_training_loop
from .metrics import _log_metric
from .ppo import _advance_noise, _should_run, minibatch_indices
from .rollout import _as_bool, tree_where
from .types import DistillationState, DistillationTransition, LoggingLevel


def default_distillation_config() -> DistillationTrainConfig:
    return DistillationTrainConfig()


def distillation_single_transition(env, teacher: StatefulModule, student: StatefulModule,
                                   carry, rng_keys_for_env_reset, reset_states=None):
    """distillation.py:66-118: both networks see the observation, the student's action
    steps the env, the teacher's `rollout_extras` are kept as the target."""
    student_state, teacher_state, env_state = carry
    student_out = student(student_state, env_state.obs)
    teacher_out = teacher(teacher_state, env_state.obs)
    student_output = student_out.output
    reset_env_state = None
    step_and_reset = getattr(env, "step_and_reset", None) if reset_states is not None else None
    if step_and_reset is not None:  # step + reset-on-done select in one launch
        next_env_state, reset_env_state = step_and_reset(env_state, student_output.actions,
                                                         reset_states)
    else:
        next_env_state = env.step(env_state, student_output.actions)
    done = _as_bool(next_env_state.done)
    trunc = next_env_state.info.get("truncated", None)
    trunc = torch.zeros_like(done) if trunc is None else _as_bool(trunc)
    transition = DistillationTransition(
        obs=env_state.obs,
        student_output=student_output,
        rewards=next_env_state.reward,
        done=done,
        truncated=trunc,
        next_obs=next_env_state.obs,
        metrics={"env": next_env_state.metrics, "student": student_out.metrics},
        student_rollout_extras=student_out.rollout_extras,
        teacher_rollout_extras=teacher_out.rollout_extras,
    )
    if reset_env_state is not None:
        next_env_state = reset_env_state
    else:
        if reset_states is None:
            reset_states = env.reset(rng_keys_for_env_reset)
        next_env_state = tree_where(done, reset_states, next_env_state)
    next_student_state = tree_where(done, student.reset_state(student_out.next_state),
                                    student_out.next_state)
    next_teacher_state = tree_where(done, teacher.reset_state(teacher_out.next_state),
                                    teacher_out.next_state)
    return (next_student_state, next_teacher_state, next_env_state), transition


def distillation_unroll_env(env, env_state, teacher: StatefulModule, student: StatefulModule,
                            student_state, teacher_state, unroll_length: int,
                            rng_key_for_env_reset: torch.Tensor):
    """distillation.py:121-157 — returns (final_student_state, final_teacher_state,
    final_env_state, DistillationTransition with `[T, N, ...]` leaves)."""
    batch_size = env_state.done.shape[0]
    keys = rnd.split(rng_key_for_env_reset, (unroll_length, batch_size))
    # every step's reset state in one batched call, as `rollout.unroll_env` does
    TN = unroll_length * batch_size
    resets = env.reset(keys.reshape(TN))

    def at_step(t):
        return tree_map(
            lambda x: x.view(unroll_length, batch_size, *x.shape[1:])[t]
            if isinstance(x, torch.Tensor) and x.dim() >= 1 and x.shape[0] == TN else x,
            resets)

    carry = (student_state, teacher_state, env_state)
    steps = []
    for t in range(unroll_length):
        carry, tr = distillation_single_transition(env, teacher, student, carry, keys[t],
                                                   reset_states=at_step(t))
        steps.append(tr)
    rollout_data = rollout.stack_steps(steps)
    return carry[0], carry[1], carry[2], rollout_data


def distillation_loss(student: StatefulModule, student_state: Any,
                      rollout_data: DistillationTransition, logging_level: LoggingLevel, *,
                      backward: bool = True) -> tuple[torch.Tensor, dict]:
    """distillation.py:160-230.  Replays the student over the trajectory with the
    TEACHER's `rollout_extras` as the replay channel, so every sampler scores the
    teacher's action mean; loss = sum over heads of -mean(log-likelihood) + sum of
    mean(regularisation).  Where the reference returns gradients from `nnx.grad`
    this ACCUMULATES them into `Parameter.grad` (as `ppo.ppo_loss` does)."""
    done = rollout_data.done
    T, B = done.shape
    ctx, out, reg_seq, _ = student.replay(student_state, rollout_data.obs, done,
                                          rollout_data.teacher_rollout_extras,
                                          need_input_grad=False)
    lls = [ll for ll in tree_leaves(out.loglikelihoods) if isinstance(ll, torch.Tensor)]
    if not lls:
        raise ValueError("distillation_loss: the student produced no log-likelihoods")
    nll_loss = -lls[0].mean()
    for ll in lls[1:]:
        nll_loss = nll_loss - ll.mean()
    regs = [r for r in tree_leaves(reg_seq) if isinstance(r, torch.Tensor)]
    regularization_loss = torch.zeros((), dtype=torch.float32, device=done.device)
    for r in regs:
        regularization_loss = regularization_loss + r.mean()
    total_loss = nll_loss + regularization_loss
    if backward:
        # d(-mean ll)/d ll = -1/(T B) for every head; the value port gets no gradient
        g_ll = tree_map(lambda ll: torch.full_like(ll, -1.0 / ll.numel()), out.loglikelihoods)
        g_v = tree_map(lambda v: torch.zeros_like(v), out.value_estimates)
        student.replay_backward(
            ctx, PPONetworkOutput(actions=None, loglikelihoods=g_ll, value_estimates=g_v),
            1.0 / float(T * B))
    loss_metrics: dict = {}
    if LoggingLevel.LOSSES in logging_level:
        loss_metrics["losses/distillation_nll"] = nll_loss
        loss_metrics["losses/regularization"] = regularization_loss
    return total_loss, loss_metrics


def distillation_step(env, teacher: StatefulModule, distillation_state: DistillationState,
                      n_envs: int, rollout_length: int, n_epochs: int, n_minibatches: int,
                      logging_level: LoggingLevel = LoggingLevel.LOSSES,
                      logging_percentiles: Optional[tuple] = None, *,
                      minibatch_inds: Optional[torch.Tensor] = None
                      ) -> tuple[DistillationState, dict]:
    """distillation.py:233-364: rollout with both networks, then
    `n_epochs * n_minibatches` student updates, then the student's statistics update.
    `minibatch_inds` injects the minibatch indices (parity tests)."""
    student: StatefulModule = distillation_state.student
    optimizer: Optimizer = distillation_state.optimizer
    bump_param_epoch()
    keys = rnd.split(distillation_state.rng_key)
    reset_key, new_key = keys[0], keys[1]
    total_iterations = n_epochs * n_minibatches
    all_indices = (minibatch_indices(new_key, n_envs, n_epochs, n_minibatches)
                   if minibatch_inds is None else minibatch_inds)
    assert all_indices.shape[0] == total_iterations

    next_student_state, next_teacher_state, next_env_state, rollout_data = (
        distillation_unroll_env(env, distillation_state.env_states, teacher, student,
                                distillation_state.student_states,
                                distillation_state.teacher_states, rollout_length, reset_key))

    # only the fields the loss reads are gathered (distillation.py:297 gathers every leaf)
    loss_view = DistillationTransition(
        obs=rollout_data.obs,
        student_output=PPONetworkOutput(actions=None, loglikelihoods=None,
                                        value_estimates=None),
        rewards=None, done=rollout_data.done, truncated=None, next_obs=None, metrics={},
        student_rollout_extras=None,
        teacher_rollout_extras=rollout_data.teacher_rollout_extras)

    per_step: dict = {}
    # every minibatch of the update in one gather launch (as ppo_step does)
    mb_leaves = tree_leaves(loss_view)
    st_leaves = tree_leaves(distillation_state.student_states)
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
        student_state_subset = tree_map(lambda x: next(it).squeeze(0),
                                        distillation_state.student_states)
        optimizer.begin(defer_dw=True)
        _, lm = distillation_loss(student, student_state_subset, minibatch, logging_level)
        for k, v in lm.items():
            per_step.setdefault(k, []).append(v)
        optimizer.update(have_norm=False)

    total_steps = distillation_state.steps_taken + rollout_length * n_envs
    student.update_statistics(rollout_data.student_rollout_extras)
    _advance_noise(student)
    _advance_noise(teacher)

    metrics: dict = {}
    for k, v in per_step.items():  # stacked over the gradient steps, as the scan does
        _log_metric(metrics, k, torch.stack(v, dim=0), logging_percentiles)
    if LoggingLevel.TRAIN_ROLLOUT_STATS in logging_level:
        _log_metric(metrics, "rollout_batch/reward", rollout_data.rewards, logging_percentiles)
        _log_metric(metrics, "rollout_batch/action", rollout_data.student_output.actions,
                    logging_percentiles)
        metrics["rollout_batch/done_rate"] = rollout_data.done.float().mean()
        metrics["rollout_batch/truncation_rate"] = rollout_data.truncated.float().mean()
    if LoggingLevel.TRAINING_ENV_METRICS in logging_level:
        for k, v in rollout_data.metrics.items():
            _log_metric(metrics, k, v, logging_percentiles)
    metrics["total_steps"] = total_steps

    distillation_state = distillation_state.replace(
        student_states=next_student_state,
        teacher_states=next_teacher_state,
        env_states=next_env_state,
        rng_key=new_key,
        steps_taken=total_steps,
    )
    return distillation_state, metrics


def new_distillation_state(env, teacher: StatefulModule, student: StatefulModule, n_envs: int,
                           seed: int, learning_rate: float = 1e-4,
                           gradient_clipping: Optional[float] = None,
                           weight_decay: Optional[float] = None, *,
                           device=None) -> DistillationState:
    """distillation.py:367-417.  Moves both networks to the device; the optimiser
    tracks the student's parameters only."""
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError(
                "nnx_ppo_amd needs a GPU: the distillation path runs on HIP kernels and has "
                "no CPU fallback")
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    key = rnd.key(seed, device)
    ks = rnd.split(key)
    key, training_key = ks[0], ks[1]
    env_init_keys = rnd.split(key, n_envs)
    env_states = tree_map(lambda x: x.clone() if isinstance(x, torch.Tensor) else x,
                          env.reset(env_init_keys))
    student.to(device)
    teacher.to(device)
    student_states = student.initialize_state(n_envs)
    teacher_states = teacher.initialize_state(n_envs)
    optimizer = Optimizer(student, learning_rate, gradient_clipping, weight_decay,
                          device=device)
    return DistillationState(student, student_states, teacher_states, env_states, optimizer,
                             training_key, torch.zeros((), dtype=torch.int64, device=device))


def train_distillation(
    env,
    teacher: StatefulModule,
    student: StatefulModule,
    config: Optional[DistillationTrainConfig] = None,
    *,
    total_steps: Optional[int] = None,
    seed: Optional[int] = None,
    log_fn: Optional[Callable[[dict, int], None]] = None,
    video_fn: Optional[Callable[[VideoData], None]] = None,
    checkpoint_fn: Optional[Callable[[DistillationState, int], None]] = None,
    eval_env=None,
    initial_state: Optional[DistillationState] = None,
    compute_dtype: Optional[str] = None,
    hip_graph: Optional[bool] = None,
    overlap_logging: Optional[bool] = None,
) -> DistillationTrainResult:
    """distillation.py:420-603.  The teacher is put in eval (deterministic) mode, so its
    samplers emit their mean; the student is trained in place.  `compute_dtype`,
    `hip_graph`, `overlap_logging`: as in `train_ppo` (they override `config.backend`)."""
    if config is None:
        config = default_distillation_config()
    if total_steps is not None:
        config = dataclasses.replace(
            config, distillation=dataclasses.replace(config.distillation,
                                                     total_steps=total_steps))
    if seed is not None:
        config = dataclasses.replace(config, seed=seed)
    if eval_env is None:
        eval_env = env
    backend = config.backend
    compute_dtype = backend.compute_dtype if compute_dtype is None else compute_dtype
    hip_graph = backend.hip_graph if hip_graph is None else hip_graph
    overlap_logging = backend.overlap_logging if overlap_logging is None else overlap_logging
    dtype_ctx = (mi_config.use_compute_dtype(compute_dtype) if compute_dtype is not None
                 else contextlib.nullcontext())
    with dtype_ctx:
        return _train_distillation(env, teacher, student, config, log_fn, checkpoint_fn,
                                   eval_env, initial_state, bool(hip_graph),
                                   bool(overlap_logging))


def _train_distillation(env, teacher, student, config, log_fn, checkpoint_fn, eval_env,
                        initial_state, hip_graph: bool,
                        overlap: bool) -> DistillationTrainResult:
    teacher.eval()
    dc = config.distillation
    if initial_state is None:
        distillation_state = new_distillation_state(
            env, teacher, student, dc.n_envs, config.seed, dc.learning_rate,
            dc.gradient_clipping, dc.weight_decay)
    else:
        distillation_state = initial_state
    device = distillation_state.steps_taken.device

    eval_history: list[dict] = []
    last_eval_step = -config.eval.every_steps
    last_checkpoint_step = -config.checkpoint_every_steps
    metrics: dict = {}

    def run_eval(steps: int) -> dict:
        student.eval()
        eval_metrics = rollout.eval_rollout(
            eval_env, student, config.eval.n_envs, config.eval.max_episode_length,
            rnd.key(config.seed, device), config.eval.logging_percentiles)
        student.train()
        return dict(eval_metrics)

    # video rendering needs a MuJoCo renderer (rollout.py:150-267): out of scope, as in
    # train_ppo — `video_fn` is accepted and never called
    steps = int(distillation_state.steps_taken)
    if config.eval.enabled:
        eval_metrics = run_eval(steps)
        metrics.update(eval_metrics)
        eval_history.append({"step": steps, **eval_metrics})
        last_eval_step = steps
    if checkpoint_fn is not None and _should_run(steps, last_checkpoint_step,
                                                 config.checkpoint_every_steps):
        checkpoint_fn(distillation_state, steps)
        last_checkpoint_step = steps
    if log_fn is not None and metrics:
        log_fn(metrics, steps)

    step_fn = lambda st: distillation_step(
        env, teacher, st, dc.n_envs, dc.rollout_length, dc.n_epochs, dc.n_minibatches,
        dc.logging_level, dc.logging_percentiles)
    runner = IterationRunner(step_fn, distillation_state, hip_graph=hip_graph,
                             networks=[student, teacher])
    loop_metrics, steps, n_iterations = run_training_loop(
        runner, total_steps=dc.total_steps, steps=steps,
        steps_per_iteration=dc.rollout_length * dc.n_envs,  # host mirror of `steps_taken`
        local_steps_per_iteration=dc.rollout_length * dc.n_envs,
        measure_throughput=LoggingLevel.THROUGHPUT in dc.logging_level,
        eval_every=config.eval.every_steps, video_every=0,
        checkpoint_every=config.checkpoint_every_steps,
        last_eval_step=last_eval_step, last_video_step=0,
        last_checkpoint_step=last_checkpoint_step,
        run_eval=run_eval if config.eval.enabled else None, run_video=None,
        checkpoint_fn=checkpoint_fn, log_fn=log_fn, eval_history=eval_history,
        overlap=overlap)
    if n_iterations:
        metrics = loop_metrics
    distillation_state = runner.state
    device_steps = int(distillation_state.steps_taken)
    if device_steps != steps:  # the host counted the iterations; the device counted the steps
        raise RuntimeError(f"the device's step counter ({device_steps}) differs from the "
                           f"host's count ({steps}): an iteration was lost or ran twice")
    return DistillationTrainResult(
        training_state=distillation_state,
        final_metrics=metrics,
        eval_history=eval_history,
        total_steps=device_steps,
        total_iterations=n_iterations,
    )
