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
# graph.  `enable_oneshot()` switches after verifying the peer path against RCCL.
_transport = "rccl"


def transport() -> str:
    return _transport if is_distributed() else "none"


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
        _collective(lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM))
        flat.mul_(1.0 / dist.get_world_size())
    return flat


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if is_distributed():
        _collective(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM))
    return t


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
    parts = [torch.empty_like(stats) for _ in range(dist.get_world_size())]
    src = stats.contiguous()
    _collective(lambda: dist.all_gather(parts, src))
    return chan_merge(parts).contiguous()
