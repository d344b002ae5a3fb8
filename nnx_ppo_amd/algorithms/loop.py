"""The host side of one training iteration, shared by `train_ppo` and
`train_distillation`: what `nnx.jit(step)` + the per-iteration host read are in the
reference (`nnx_ppo/algorithms/ppo.py:105,189-214`; `distillation.py:470-520`).

  * `IterationRunner.launch()` enqueues one iteration — the first one eagerly (it is a real
    iteration: it creates workspaces, bf16 images and lazy state), every later one as ONE
    replay of the HIP graph recorded from the second (`graph.GraphedStep`; a sharded run
    over RCCL records `graph.SegmentedStep`) — followed by ONE device-to-host copy of the
    iteration's packed metric scalars into pinned memory, and returns a ticket.
  * `IterationRunner.collect(ticket)` waits for that copy (the iteration's single host
    sync, the counterpart of `int(training_state.steps_taken)`, ppo.py:209) and returns the
    metrics as 0-d CPU tensors.
  * `run_training_loop` keeps the reference's callback cadence.  When nothing is due
    between two iterations (no eval, video or checkpoint) iteration i+1 is enqueued BEFORE
    the host waits for iteration i's metrics, so the GPU never idles on the host; the log
    callback of iteration i then runs while iteration i+1 computes.  `overlap=False`
    restores the strict order launch → read → callbacks → launch.
"""
from __future__ import annotations

import time
from typing import Any, Callable, Optional

import torch

from .. import ops, parallel
from .graph import GraphCaptureError, GraphedStep, SegmentedStep  # noqa: F401


class MetricPack:
    """Packs a dict of device scalars into one byte buffer (8-byte cells, ONE
    `mi_copy_multi` launch) so that an iteration's metrics reach the host in one copy.
    Built once per runner from the first metrics dict; the set of keys, their dtypes and
    the buffer are then fixed (a captured graph bakes the pointers in)."""

    CELL = 8

    def __init__(self, metrics: dict, device):
        self.names: list[str] = []
        self.dtypes: list[torch.dtype] = []
        self.static: dict[str, Any] = {}
        for k, v in metrics.items():
            if isinstance(v, torch.Tensor) and v.numel() == 1:
                self.names.append(k)
                self.dtypes.append(v.dtype)
            else:
                self.static[k] = v  # python numbers / non-scalar diagnostics pass through
        n = max(len(self.names), 1)
        self.buf = torch.zeros(n * self.CELL, dtype=torch.uint8, device=device)
        self.cells = [self.buf[i * self.CELL:i * self.CELL + torch.empty((), dtype=dt).element_size()]
                      .view(dt) for i, dt in enumerate(self.dtypes)]
        # two pinned landing slots: iteration i+1's copy may be enqueued before the host
        # has read iteration i's
        self.host = [torch.zeros(n * self.CELL, dtype=torch.uint8).pin_memory()
                     if device.type == "cuda" else torch.zeros(n * self.CELL, dtype=torch.uint8)
                     for _ in range(2)]

    def pack(self, metrics: dict) -> None:
        """Enqueue the copy of every scalar into its cell (capturable)."""
        pairs = []
        for k, dt, cell in zip(self.names, self.dtypes, self.cells):
            v = metrics[k]
            if v.dtype != dt:
                raise RuntimeError(f"metric {k!r} changed dtype ({dt} -> {v.dtype})")
            v = v.reshape(1)
            pairs.append((cell, v if v.is_contiguous() else v.contiguous()))
        if set(k for k, v in metrics.items()
               if isinstance(v, torch.Tensor) and v.numel() == 1) != set(self.names):
            raise RuntimeError("the set of metric keys changed between iterations")
        if pairs:
            ops.copy_multi(pairs)

    def to_host(self, slot: int) -> None:
        self.host[slot].copy_(self.buf, non_blocking=True)

    def decode(self, slot: int) -> dict:
        raw = self.host[slot].clone()  # detach from the pinned slot before it is reused
        out = dict(self.static)
        for i, (k, dt) in enumerate(zip(self.names, self.dtypes)):
            out[k] = raw[i * self.CELL:i * self.CELL + torch.empty((), dtype=dt).element_size()] \
                .view(dt).reshape(())
        return out


_HEALTH_KEY = "_health/"


class InKernelTimeout(RuntimeError):
    """A bounded wait inside a kernel ran out during this iteration."""


class IterationRunner:
    """Runs `fn(state) -> (state, metrics)` once per `launch()`.

    hip_graph: True — iteration 1 eager, iteration 2 recorded, later ones replayed; a
               capture failure raises `GraphCaptureError` (never a silent eager run);
               False — every iteration launches its kernels from Python.
    """

    def __init__(self, fn: Callable, state, *, hip_graph: bool = True, networks=()):
        self.fn = fn
        self._state = state
        self.hip_graph = bool(hip_graph)
        # modules whose samplers may be called BETWEEN iterations (eval rollouts, user
        # code): each such call takes a noise offset that an eager iteration folds into
        # the device counter at its end, but a replayed graph has its own count baked in —
        # so pending offsets are flushed before every replay (same noise either way)
        self._networks = list(networks)
        self.device = state.steps_taken.device
        # ONE stream for every launch of the run (eager iteration, capture, replays, metric
        # copies): per-stream workspaces made by the eager iteration are the ones the
        # captured launches use, and a capture needs a non-default stream anyway.  Each
        # launch first waits for the caller's current stream (eval / checkpoint work).
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self._graph = None
        self._pack: Optional[MetricPack] = None
        self._health: list = []
        self._n = 0
        self._events: list = [None, None]
        self.launch_mode = "eager"

    @property
    def state(self):
        return self._state

    def _after(self, metrics: dict) -> None:
        if self._pack is None:
            # the sticky timeout words of every bounded in-kernel wait that exists after the
            # first (eager) iteration ride in the same copy as the metrics: `collect` raises
            # in the iteration one of them turns non-zero
            self._health = ops.health_words(self.device) if self.device.type == "cuda" else []
        metrics = dict(metrics)
        for i, w in enumerate(self._health):
            metrics[f"{_HEALTH_KEY}{i}"] = w
        if self._pack is None:
            self._pack = MetricPack(metrics, self.device)
        self._pack.pack(metrics)

    def launch(self) -> int:
        """Enqueue one iteration and the copy of its metrics; returns the ticket."""
        if self.stream is None:
            return self._launch()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            return self._launch()

    def _launch(self) -> int:
        if self.hip_graph and self._n >= 1:
            for net in self._networks:  # before the capture, and before every replay
                for m in net.modules():
                    if getattr(m, "_pending", 0) and hasattr(m, "advance_rng"):
                        m.advance_rng()
            if self._graph is None:
                # the static buffers of the graph: own copies of every state leaf (a leaf
                # may alias a shared read-only constant an env hands out)
                from ..tree import tree_map
                from .graph import _tensor_fields

                own = lambda x: x.clone() if isinstance(x, torch.Tensor) else x
                self._state = self._state.replace(**{
                    n: tree_map(own, getattr(self._state, n))
                    for n in _tensor_fields(self._state)})
                # sharded over RCCL: the collectives cannot live inside a capture
                # (graph.py docstring) — a sequence of graphs with eager collectives
                # between them; otherwise (one GPU, or the one-shot peer kernels) ONE graph
                segmented = parallel.is_distributed() and parallel.transport() == "rccl"
                Recorder = SegmentedStep if segmented else GraphedStep
                self._graph = Recorder(self.fn, self._state, warmup=0, after=self._after,
                                       stream=self.stream)
                self._state = self._graph.ts
                self.launch_mode = (
                    f"{sum(isinstance(x, torch.cuda.CUDAGraph) for x in self._graph.program)} "
                    "HIP graphs per iteration with eager RCCL collectives between them"
                    if segmented else "hip-graph (one hipGraphLaunch per iteration)")
            self._graph()
        else:
            self._state, metrics = self.fn(self._state)
            self._after(metrics)
        slot = self._n & 1
        self._pack.to_host(slot)
        if self.device.type == "cuda":
            ev = self._events[slot]
            if ev is None:
                ev = self._events[slot] = torch.cuda.Event()
            ev.record()
        self._n += 1
        return self._n - 1

    def collect(self, ticket: int) -> dict:
        """Wait for iteration `ticket`'s metrics (its one host sync) and return them."""
        slot = ticket & 1
        if ticket < self._n - 2:
            raise RuntimeError("metrics of an iteration older than the previous one are gone")
        if self.device.type == "cuda":
            self._events[slot].synchronize()
        metrics = self._pack.decode(slot)
        bad = sum(int(metrics.pop(k)) for k in list(metrics) if k.startswith(_HEALTH_KEY))
        if bad:
            raise InKernelTimeout(
                f"iteration {ticket}: {bad} bounded in-kernel wait(s) ran out (an advantage-"
                "statistics hand-over between the workgroups of mi_policy_ws_bwd_gae_bf16 / "
                "mi_gae_ppo_loss_f32, or a peer that never delivered its chunk to a one-shot "
                "exchange).  The step that saw it was poisoned (NaN statistics / no optimiser "
                "update), nothing after it can be trusted: the run stops here")
        return metrics


def should_run(steps: int, last_step: int, every_steps: int) -> bool:
    """ppo.py:34-38."""
    if every_steps <= 0:
        return False
    return (steps // every_steps) > (last_step // every_steps)


def health_check(device) -> None:
    """Raise if a bounded in-kernel wait has run out since the last check: a peer that never
    delivered its chunks to a one-shot exchange (`comm.PeerComm.check`), or a workgroup that
    never published its advantage-statistics partial.  `IterationRunner.collect` already sees
    the words that existed when its metric pack was built, every iteration; this reads EVERY
    word (it synchronises), so the loop calls it where it is synchronised anyway — BEFORE an
    eval, a video or a checkpoint may use the parameters, and at the end of training."""
    from .. import ops, parallel

    del device
    comm = parallel.peer_comm()
    if comm is not None:
        comm.check()
    n = ops.handover_timeouts()
    if n:
        raise InKernelTimeout(
            f"{n} advantage-statistics hand-over(s) inside mi_policy_ws_bwd_gae_bf16 / "
            "mi_gae_ppo_loss_f32 timed out: a workgroup of the launch was not resident within "
            "2 s (another kernel holding the CUs?); MIPPO_GAE_IN_BWD=0 keeps the GAE / loss "
            "launch of its own")


def run_training_loop(
    runner: IterationRunner,
    *,
    total_steps: int,
    steps: int,
    steps_per_iteration: int,
    local_steps_per_iteration: int,
    measure_throughput: bool,
    eval_every: int,
    video_every: int,
    checkpoint_every: int,
    last_eval_step: int,
    last_video_step: int,
    last_checkpoint_step: int,
    run_eval: Optional[Callable[[int], dict]],
    run_video: Optional[Callable[[int, int], dict]],
    checkpoint_fn: Optional[Callable],
    log_fn: Optional[Callable[[dict, int], None]],
    eval_history: list,
    overlap: bool = True,
) -> tuple[dict, int, int]:
    """The `while steps < total_steps` loop of ppo.py:189-242 / distillation.py:520-590.
    Returns (last metrics, steps, iterations).  `steps_taken` advances by a constant, so
    the host counts it (and `train_*` checks it against the device counter once, at the
    end) instead of reading it every iteration."""
    metrics: dict = {}
    n_iterations = 0
    if steps >= total_steps:
        return metrics, steps, n_iterations
    t_prev = time.perf_counter()
    ticket = runner.launch()
    while True:
        steps += steps_per_iteration
        n_iterations += 1
        eval_due = run_eval is not None and should_run(steps, last_eval_step, eval_every)
        video_due = run_video is not None and should_run(steps, last_video_step, video_every)
        ckpt_due = checkpoint_fn is not None and should_run(steps, last_checkpoint_step,
                                                            checkpoint_every)
        more = steps < total_steps
        next_ticket = None
        if overlap and more and not (eval_due or video_due or ckpt_due):
            next_ticket = runner.launch()  # iteration i+1 queued behind iteration i
        metrics = runner.collect(ticket)   # the host sync of iteration i (raises on a timeout)
        # the host's count, as a 0-d tensor like every other logged scalar (the device
        # counter is checked against it once, at the end of train_*)
        metrics["total_steps"] = torch.tensor(steps, dtype=torch.int64)
        if measure_throughput:
            now = time.perf_counter()
            metrics["throughput/train_sps"] = local_steps_per_iteration / (now - t_prev)
        if eval_due or video_due or ckpt_due or not more:
            # before anything evaluates, renders or SAVES these parameters
            health_check(runner.device)
        if eval_due:
            eval_metrics = run_eval(steps)
            metrics.update(eval_metrics)
            eval_history.append({"step": steps, **eval_metrics})
            last_eval_step = steps
        if video_due:
            metrics.update(run_video(steps, n_iterations))
            last_video_step = steps
        if ckpt_due:
            checkpoint_fn(runner.state, steps)
            last_checkpoint_step = steps
        if log_fn is not None:
            log_fn(metrics, steps)
        if not more:
            break
        if (eval_due or video_due or ckpt_due) and parallel.transport() == "oneshot":
            # callbacks often do rank-0-only work (rendering, uploads, file writes); the
            # one-shot exchange waits for a peer only `MIPPO_COMM_TIMEOUT_S` inside a kernel,
            # so the ranks meet on the host before any of them launches into it again
            parallel.host_barrier()
        # the clock of the next iteration starts where this one's was read, so callback
        # time that the GPU spent computing the next iteration is not counted twice
        t_prev = time.perf_counter() if (eval_due or video_due or ckpt_due or not overlap) \
            else now if measure_throughput else t_prev
        ticket = next_ticket if next_ticket is not None else runner.launch()
    return metrics, steps, n_iterations
