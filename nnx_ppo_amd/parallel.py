"""Env-sharded data parallelism: one process per GPU, each owning `n_envs`
environments end to end; the only exchanges are small all-reduces over RCCL
(`torch.distributed`, backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).

The reference is single-device (no pmap / mesh / collective anywhere), so every
collective here is new:
  (1) gradient all-reduce-mean of the flat gradient arena, once per grad step;
  (2) all-reduce of the advantage statistics (sum, sum of squares, count) so the
      minibatch normalisation of `ppo.py:477-480` is over the GLOBAL minibatch;
  (3) Chan merge of per-shard normaliser batch statistics, once per iteration,
      in rank order, so every replica holds bit-identical statistics;
  (4) loss scalars averaged for logging.
Without an initialised process group every function is the identity.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size() -> int:
    return dist.get_world_size() if is_distributed() else 1


def rank() -> int:
    return dist.get_rank() if is_distributed() else 0


# How the exchanges travel: "rccl" = torch.distributed collectives (RCCL on GPUs, gloo in
# the CPU tests); "oneshot" = the one-shot peer kernels over IPC-mapped buffers
# (`comm.py`, csrc/comm.hip) — plain kernel launches, so a sharded iteration is ONE HIP
# graph.  `enable_oneshot()` switches after verifying the peer path against
# torch.distributed on seeded data; every rank takes the same decision.
_transport = "rccl"
_comm = None


def transport() -> str:
    return _transport if is_distributed() else "none"


def peer_comm():
    return _comm if (is_distributed() and _transport == "oneshot") else None


def enable_oneshot(device, slot_bytes: int | None = None,
                   timeout_s: float | None = None) -> tuple[bool, str]:
    """Collective.  Build the peer communicator, check it against torch.distributed and
    switch the exchanges to it.  Returns (enabled, reason-if-not)."""
    global _transport, _comm
    if not is_distributed():
        return False, "no process group"
    from . import comm as comm_mod

    gpu_backend = dist.get_backend() == "nccl"
    why = None
    c = None
    try:
        kw = {}
        if slot_bytes is not None:
            kw["slot_bytes"] = slot_bytes
        if timeout_s is not None:
            kw["timeout_s"] = timeout_s
        c = comm_mod.PeerComm(device, **kw)
    except Exception as exc:  # noqa: BLE001 - the reason it is not used
        why = f"{type(exc).__name__}: {exc}"
    # every rank must have a communicator before any of them launches into it
    ok = torch.tensor([1 if c is not None else 0], dtype=torch.int32)
    ok = ok.to(device) if gpu_backend else ok
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        if c is not None:
            c.close()
        return False, why or "a peer rank could not create its communicator"
    why = comm_mod.self_check(c, device, gpu_backend)
    if why is not None:
        c.close()
        return False, why
    _comm, _transport = c, "oneshot"
    return True, ""


def disable_oneshot() -> None:
    global _transport, _comm
    if _comm is not None:
        _comm.close()
    _comm, _transport = None, "rccl"


# While an iteration is being recorded as a SEQUENCE of HIP graphs
# (algorithms/graph.py:SegmentedPPOStep) the recorder sits here: every collective
# closes the graph being captured, runs eagerly, and opens the next one.
_segmenter = None


def _collective(fn) -> None:
    if _segmenter is not None:
        _segmenter.collective(fn)
    else:
        fn()


def allreduce_mean_(flat: torch.Tensor) -> torch.Tensor:
    """In-place mean over ranks (gradient arena; equal shard sizes, so the mean of
    per-shard mean-losses' gradients is the global-minibatch gradient)."""
    if is_distributed():
        c = peer_comm()
        if c is not None and flat.is_cuda and flat.dtype in (torch.float32, torch.float64):
            return c.allreduce_(flat, 1.0 / dist.get_world_size())
        _collective(lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM))
        flat.mul_(1.0 / dist.get_world_size())
    return flat


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if is_distributed():
        c = peer_comm()
        if c is not None and t.is_cuda and t.is_contiguous() and \
                t.dtype in (torch.float32, torch.float64):
            return c.allreduce_(t, 1.0)
        _collective(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM))
    return t


def host_barrier() -> None:
    """Host-side rendezvous of all ranks (no-op without a process group)."""
    if is_distributed():
        dist.barrier()


def chan_merge(stats_list: list[torch.Tensor]) -> torch.Tensor:
    """Merge per-shard `[3, F]` = (n, mean, M2) statistics pairwise in list order
    (Chan et al.; the same formula as `normalizer.py:118-133`)."""
    n, mean, m2 = stats_list[0][0].clone(), stats_list[0][1].clone(), stats_list[0][2].clone()
    for s in stats_list[1:]:
        nb, mb, m2b = s[0], s[1], s[2]
        tot = n + nb
        delta = mb - mean
        mean = mean + delta * (nb / tot)
        m2 = m2 + m2b + delta * delta * (n * nb / tot)
        n = tot
    return torch.stack([n, mean, m2])


def merge_batch_stats(stats: torch.Tensor) -> torch.Tensor:
    """All-gather per-shard normaliser batch statistics and merge in rank order."""
    if not is_distributed():
        return stats
    src = stats.contiguous()
    c = peer_comm()
    if c is not None and src.is_cuda:
        return chan_merge(list(c.allgather(src).unbind(0))).contiguous()
    parts = [torch.empty_like(stats) for _ in range(dist.get_world_size())]
    _collective(lambda: dist.all_gather(parts, src))
    return chan_merge(parts).contiguous()
