"""Whole-iteration HIP-graph capture of `ppo_step`.

The reference compiles one XLA program per iteration with `nnx.jit(ppo_step)`
(`nnx_ppo/algorithms/ppo.py:105,192-207`) and crosses the host/device boundary
once per iteration.  The MI355X-native equivalent is not a tracing compiler but
a captured HIP graph: one eager pass records every kernel launch of the
iteration (rollout, minibatch permutations, 16 gradient steps, normaliser
update, RNG advance) and each later iteration is ONE `hipGraphLaunch`.  At this
workload's size the iteration is launch-bound (~1500 launches of tiny kernels),
so removing the per-launch host cost is the first-order win.

What makes the iteration capturable (and is required of user envs/modules):
  * every kernel is enqueued on torch's current stream with caller-owned
    buffers; nothing synchronises or reads a device value on the host;
  * all iteration-to-iteration state is device-resident and updated in place:
    parameters / Adam moments / step counter, normaliser statistics, the
    sampler's {seed, offset} (so replays draw fresh noise), and the training
    state's env / carry / key tensors, which are copied back into the static
    input buffers at the end of the captured region.
"""
from __future__ import annotations

from typing import Any

import torch

from ..tree import tree_leaves
from .ppo import ppo_step
from .types import TrainingState


def _copy_state(dst: TrainingState, src: TrainingState) -> None:
    pairs = []
    for name in ("network_states", "env_states", "rng_key", "steps_taken"):
        d = tree_leaves(getattr(dst, name))
        s = tree_leaves(getattr(src, name))
        if len(d) != len(s):
            raise RuntimeError(
                f"ppo_step changed the structure of training_state.{name}; "
                "HIP-graph capture needs a stable state pytree")
        for a, b in zip(d, s):
            if not isinstance(a, torch.Tensor):
                continue
            if a.shape != b.shape or a.dtype != b.dtype:
                raise RuntimeError(
                    f"training_state.{name}: leaf changed from {a.dtype}{tuple(a.shape)} to "
                    f"{b.dtype}{tuple(b.shape)}; HIP-graph capture needs stable leaves")
            if a.data_ptr() != b.data_ptr():
                if a.is_cuda and a.is_contiguous() and b.is_contiguous():
                    pairs.append((a, b))
                else:
                    a.copy_(b)
    if pairs:  # one launch for all leaves
        from .. import ops

        ops.copy_multi(pairs)


class GraphedPPOStep:
    """`step = GraphedPPOStep(env, training_state, *ppo_step_args)`; then
    `training_state, metrics = step()` runs one iteration as a single graph
    launch.  `training_state` keeps the same (static) tensors across calls;
    `metrics` are device scalars refreshed by every replay.  `warmup` eager
    iterations run first (they are real iterations and advance training)."""

    def __init__(self, env, training_state: TrainingState, *args: Any, warmup: int = 2,
                 **kwargs: Any):
        self.env = env
        self.ts = training_state
        self.args, self.kwargs = args, kwargs
        self.warmup_iterations = warmup
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                new_ts, _ = ppo_step(env, self.ts, *args, **kwargs)
                _copy_state(self.ts, new_ts)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # sharded run: the collectives are captured with the iteration; the process
        # group's watchdog thread keeps polling its own events meanwhile, which only a
        # thread-local capture mode tolerates
        from .. import parallel

        mode = "thread_local" if parallel.is_distributed() else "global"
        # Capture by hand rather than with `torch.cuda.graph`: if anything in the
        # iteration cannot be captured, the capture must still be ENDED (that is what
        # takes the streams out of capture mode) and the current stream restored, so
        # that the caller can fall back to eager launches in the same process.
        cap = torch.cuda.Stream()
        cap.wait_stream(cur)
        failure = None
        with torch.cuda.stream(cap):
            self.graph.capture_begin(capture_error_mode=mode)
            try:
                new_ts, metrics = ppo_step(env, self.ts, *args, **kwargs)
                _copy_state(self.ts, new_ts)
            except BaseException as exc:  # noqa: BLE001 - re-raised below
                failure = exc
            try:
                self.graph.capture_end()
            except Exception as exc:  # an invalidated capture reports itself here
                failure = failure or exc
        cur.wait_stream(cap)
        if failure is not None:
            try:
                torch.cuda.synchronize()
            except Exception:  # the sticky error of the aborted capture
                pass
            # streams that had been forked into the aborted capture do not accept
            # launches any more: let the port-overlap code create fresh ones
            from ..networks import adapter

            adapter._SIDE_STREAMS.clear()
            raise RuntimeError(f"HIP-graph capture of ppo_step failed: {failure!r}") from failure
        self.metrics = metrics

    def __call__(self):
        self.graph.replay()
        return self.ts, self.metrics


class SegmentedPPOStep:
    """A sharded iteration as a SEQUENCE of HIP graphs with the collectives between them.

    Capturing RCCL calls into a HIP graph is the fastest form (`GraphedPPOStep` in a
    sharded run), but an aborted capture leaves the process unusable and a wedged
    replay cannot be recovered from.  This recorder needs nothing from the collective
    library: `parallel._collective` closes the graph under capture, runs the collective
    eagerly, and opens the next graph in the same memory pool.  An iteration of the
    BASELINE workload becomes ~35 graph launches + 34 eager collectives instead of ~500
    kernel launches from Python.  Same contract as `GraphedPPOStep`."""

    def __init__(self, env, training_state: TrainingState, *args: Any, warmup: int = 2,
                 **kwargs: Any):
        from .. import parallel

        self.env = env
        self.ts = training_state
        self.program: list = []
        cur = torch.cuda.current_stream()
        self._stream = torch.cuda.Stream()
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            for _ in range(warmup):
                new_ts, _ = ppo_step(env, self.ts, *args, **kwargs)
                _copy_state(self.ts, new_ts)
        cur.wait_stream(self._stream)
        torch.cuda.synchronize()
        self._pool = torch.cuda.graph_pool_handle()
        self._cur = None
        failure = None
        with torch.cuda.stream(self._stream):
            self._begin()
            parallel._segmenter = self
            try:
                new_ts, metrics = ppo_step(env, self.ts, *args, **kwargs)
                _copy_state(self.ts, new_ts)
            except BaseException as exc:  # noqa: BLE001 - re-raised below
                failure = exc
            finally:
                parallel._segmenter = None
            try:
                self._end()
            except Exception as exc:
                failure = failure or exc
        cur.wait_stream(self._stream)
        if failure is not None:
            raise RuntimeError(f"segmented capture of ppo_step failed: {failure!r}") from failure
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
