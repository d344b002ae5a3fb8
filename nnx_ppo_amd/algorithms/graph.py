"""Whole-iteration HIP-graph capture of a training step (`ppo_step`,
`distillation_step`).

The reference compiles one XLA program per iteration with `nnx.jit(ppo_step)`
(`nnx_ppo/algorithms/ppo.py:105,192-207`; `distillation.py:470-480`) and crosses the
host/device boundary once per iteration.  The MI355X-native equivalent is not a tracing
compiler but a captured HIP graph: one eager pass records every kernel launch of the
iteration (rollout, minibatch permutations, 16 gradient steps, normaliser update, RNG
advance) and each later iteration is ONE `hipGraphLaunch`.  At this workload's size the
iteration is launch-bound, so removing the per-launch host cost is the first-order win.

What makes the iteration capturable (and is required of user envs/modules):
  * every kernel is enqueued on torch's current stream with caller-owned
    buffers; nothing synchronises or reads a device value on the host;
  * all iteration-to-iteration state is device-resident and updated in place:
    parameters / Adam moments / step counter, normaliser statistics, the
    sampler's {seed, offset} (so replays draw fresh noise), and the training
    state's env / carry / key tensors, which are copied back into the static
    input buffers at the end of the captured region.

A capture that fails RAISES (`GraphCaptureError`): the caller decides whether to run
eagerly (`train_ppo(..., hip_graph=False)`); nothing here continues eagerly by itself.
With one process per GPU and no collective inside the capture an aborted capture leaves
the process usable (`tests/test_train_loop_gpu.py::test_capture_failure_is_clean`); with
RCCL collectives recorded INTO the graph it did not (the process group's communicator and
watchdog stay bound to the invalidated capture — see DESIGN §7), which is why a sharded
run never records RCCL calls: it uses the one-shot peer kernels (plain launches) or
`SegmentedStep`.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Callable

import torch

from ..tree import tree_leaves
from .types import TrainingState


class GraphCaptureError(RuntimeError):
    """The iteration could not be recorded into a HIP graph (something in it — usually
    the env's `step` / `reset` — synchronises, reads a device value on the host or
    allocates pinned memory).  Run with `hip_graph=False`."""


def _tensor_fields(state) -> list[str]:
    """The fields of a training-state dataclass that hold tensors (pytrees of them) —
    everything but the module / optimiser objects, which are mutated in place."""
    from ..networks.types import StatefulModule
    from ..optim import Optimizer

    return [f.name for f in dataclasses.fields(state)
            if not isinstance(getattr(state, f.name), (StatefulModule, Optimizer))]


def _copy_state(dst, src) -> None:
    pairs = []
    for name in _tensor_fields(dst):
        d = tree_leaves(getattr(dst, name))
        s = tree_leaves(getattr(src, name))
        if len(d) != len(s):
            raise RuntimeError(
                f"the training step changed the structure of state.{name}; "
                "HIP-graph capture needs a stable state pytree")
        for a, b in zip(d, s):
            if not isinstance(a, torch.Tensor):
                continue
            if a.shape != b.shape or a.dtype != b.dtype:
                raise RuntimeError(
                    f"state.{name}: leaf changed from {a.dtype}{tuple(a.shape)} to "
                    f"{b.dtype}{tuple(b.shape)}; HIP-graph capture needs stable leaves")
            if a.data_ptr() != b.data_ptr():
                if a.is_cuda and a.is_contiguous() and b.is_contiguous():
                    pairs.append((a, b))
                else:
                    a.copy_(b)
    if pairs:  # one launch for all leaves
        from .. import ops

        ops.copy_multi(pairs)


class GraphedStep:
    """`g = GraphedStep(fn, state)` with `fn(state) -> (new_state, metrics)`; then
    `state, metrics = g()` runs one iteration as a single graph launch.  `state` keeps
    the same (static) tensors across calls; `metrics` are device scalars refreshed by
    every replay.  `warmup` eager iterations run first (they are real iterations and
    advance training); `after(metrics)` (optional) is recorded at the end of the
    iteration (the training loop packs the metrics for one device-to-host copy there)."""

    def __init__(self, fn: Callable, state, *, warmup: int = 1,
                 after: Callable[[dict], Any] | None = None, stream=None):
        self.fn = fn
        self.ts = state
        self.warmup_iterations = warmup
        self.after_result = None
        cur = torch.cuda.current_stream()
        # ONE stream for warm-up and capture: per-stream workspaces (ops.workspace)
        # created by the warm-up are the ones the captured launches use.  `stream`: the
        # caller's launch stream (it may be the current one); a fresh one otherwise.
        self.stream = stream if stream is not None else torch.cuda.Stream()
        if self.stream != cur:
            self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                new_ts, m = fn(self.ts)
                _copy_state(self.ts, new_ts)
                if after is not None:
                    after(m)
        if self.stream != cur:
            cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # Capture by hand rather than with `torch.cuda.graph`: if anything in the
        # iteration cannot be captured, the capture must still be ENDED (that is what
        # takes the streams out of capture mode) before the error is reported.
        failure = None
        # sharded run: the process group's watchdog thread keeps polling its own events
        # meanwhile, which only a thread-local capture mode tolerates
        from .. import parallel

        mode = "thread_local" if parallel.is_distributed() else "global"
        with torch.cuda.stream(self.stream):
            self.graph.capture_begin(capture_error_mode=mode)
            try:
                new_ts, metrics = fn(self.ts)
                _copy_state(self.ts, new_ts)
                if after is not None:
                    self.after_result = after(metrics)
            except BaseException as exc:  # noqa: BLE001 - re-raised below
                failure = exc
            try:
                self.graph.capture_end()
            except Exception as exc:  # an invalidated capture reports itself here
                failure = failure or exc
        if self.stream != cur:
            cur.wait_stream(self.stream)
        if failure is not None:
            self.graph = None
            try:
                torch.cuda.synchronize()
            except Exception:  # the sticky error of the aborted capture, reported once
                pass
            # streams that had been forked into the aborted capture are not reused
            from ..networks import adapter

            adapter._SIDE_STREAMS.clear()
            if isinstance(failure, (KeyboardInterrupt, SystemExit)):
                raise failure
            raise GraphCaptureError(
                f"HIP-graph capture of the training step failed: {failure!r}") from failure
        self.metrics = metrics

    def __call__(self):
        self.graph.replay()
        return self.ts, self.metrics


class SegmentedStep:
    """A sharded iteration as a SEQUENCE of HIP graphs with the collectives between them.

    For the RCCL transport (`parallel.transport() == "rccl"`): `parallel._collective`
    closes the graph under capture, runs the collective eagerly, and opens the next graph
    in the same memory pool, so nothing is asked of the collective library.  An iteration
    of the BASELINE workload becomes ~35 graph launches + 34 eager collectives instead of
    ~260 kernel launches from Python.  (With the one-shot peer transport the collectives
    are plain kernels and `GraphedStep` records the whole iteration as ONE graph.)  Same
    contract as `GraphedStep`."""

    def __init__(self, fn: Callable, state, *, warmup: int = 1,
                 after: Callable[[dict], Any] | None = None, stream=None):
        from .. import parallel

        self.fn = fn
        self.ts = state
        self.program: list = []
        self.after_result = None
        cur = torch.cuda.current_stream()
        self.stream = stream if stream is not None else torch.cuda.Stream()
        if self.stream != cur:
            self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                new_ts, m = fn(self.ts)
                _copy_state(self.ts, new_ts)
                if after is not None:
                    after(m)
        if self.stream != cur:
            cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        self._pool = torch.cuda.graph_pool_handle()
        self._cur = None
        failure = None
        with torch.cuda.stream(self.stream):
            self._begin()
            parallel._segmenter = self
            try:
                new_ts, metrics = fn(self.ts)
                _copy_state(self.ts, new_ts)
                if after is not None:
                    self.after_result = after(metrics)
            except BaseException as exc:  # noqa: BLE001 - re-raised below
                failure = exc
            finally:
                parallel._segmenter = None
            try:
                self._end()
            except Exception as exc:
                failure = failure or exc
        if self.stream != cur:
            cur.wait_stream(self.stream)
        if failure is not None:
            if isinstance(failure, (KeyboardInterrupt, SystemExit)):
                raise failure
            raise GraphCaptureError(
                f"segmented capture of the training step failed: {failure!r}") from failure
        self.metrics = metrics

    def _begin(self) -> None:
        self._cur = torch.cuda.CUDAGraph()
        self._cur.capture_begin(pool=self._pool, capture_error_mode="thread_local")

    def _end(self) -> None:
        g, self._cur = self._cur, None
        g.capture_end()
        self.program.append(g)

    def collective(self, fn) -> None:
        self._end()
        fn()  # every rank takes part once while recording, too
        self.program.append(fn)
        self._begin()

    def __call__(self):
        for item in self.program:
            if isinstance(item, torch.cuda.CUDAGraph):
                item.replay()
            else:
                item()
        return self.ts, self.metrics


def _ppo_fn(env, args, kwargs):
    from .ppo import ppo_step

    return lambda ts: ppo_step(env, ts, *args, **kwargs)


class GraphedPPOStep(GraphedStep):
    """`step = GraphedPPOStep(env, training_state, *ppo_step_args)`; `training_state,
    metrics = step()` — `ppo_step` under `GraphedStep`."""

    def __init__(self, env, training_state: TrainingState, *args: Any, warmup: int = 2,
                 **kwargs: Any):
        self.env = env
        super().__init__(_ppo_fn(env, args, kwargs), training_state, warmup=warmup)


class SegmentedPPOStep(SegmentedStep):
    """`ppo_step` under `SegmentedStep`."""

    def __init__(self, env, training_state: TrainingState, *args: Any, warmup: int = 2,
                 **kwargs: Any):
        self.env = env
        super().__init__(_ppo_fn(env, args, kwargs), training_state, warmup=warmup)
