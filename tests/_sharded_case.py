"""The sharded-vs-single-process equivalence case (tests/test_comm_gpu.py): the same N envs
either in one process, or split N / world per rank with the exchanges of DESIGN §7.

What makes the two runs comparable value for value:
  * env states are built from the GLOBAL list of env keys (rank r takes its slice);
  * the envs never reset inside the rollout (MockEnv with a long episode), so the reset
    keys — which a shard draws for its own envs — are never used;
  * the policy is evaluated deterministically (`networks.eval()`) with
    entropy_weight = 0, so no noise stream (indexed by the row inside a batch) enters;
  * minibatch indices are injected: shard r uses a permutation of its own envs, the single
    process uses the union of the shards' slices, shard by shard (a global
    minibatch is the union of the local ones).
What differs is floating-point summation order only."""
import os

import torch

# MIPPO_SHARDED_CASE=big: BASELINE configs[1]'s network in bf16 at a size where a sharded rank's
# gradient step is the FOUR launches of a single GPU (N / world = 1024 envs, B = 512 per rank:
# mi_policy_ws_fwd_bf16, mi_policy_ws_bwd_gae_bf16 with the statistics exchanged inside it,
# mi_dense_bwd_dw_grouped_slabs_bf16, mi_adam_step_allreduce_f32 summing the slabs)
BIG = os.environ.get("MIPPO_SHARDED_CASE") == "big"
N, T, E, MB = (2048, 30, 1, 2) if BIG else (128, 8, 2, 2)
ARGS = (0.95, 0.99, 0.2, True, False, E, MB)


def _env():
    from nnx_ppo_amd.envs import MockEnv

    return MockEnv(5, 1, max_steps=1000)


def local_perms(world):
    """[world][E, N_local] shard-local permutations (seeded, the same in every process)."""
    g = torch.Generator().manual_seed(11)
    n_local = N // world
    return [[torch.randperm(n_local, generator=g) for _ in range(E)] for _ in range(world)]


def build_state(dev, rank, world):
    """(env, TrainingState, minibatch_inds) for rank `rank` of `world`; world == 1 is the
    single-process run over all N envs with the union minibatches."""
    from nnx_ppo_amd import random as rnd
    from nnx_ppo_amd.algorithms.types import TrainingState
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.optim import Optimizer
    from nnx_ppo_amd.tree import tree_map

    env = _env()
    hidden = ([64] * 4, [256] * 2) if BIG else ([32, 32], [64, 64])
    net = factories.make_mlp_actor_critic(5, 1, *hidden, Rngs(21), entropy_weight=0.0)
    net.eval()  # deterministic policy: no noise stream
    n_local = N // world
    all_keys = rnd.split(rnd.key(77, dev), N)
    mine = all_keys[rank * n_local:(rank + 1) * n_local].contiguous()
    env_states = tree_map(lambda x: x.clone() if isinstance(x, torch.Tensor) else x,
                          env.reset(mine))
    net.to(dev)
    ts = TrainingState(net, net.initialize_state(n_local), env_states,
                       Optimizer(net, 1e-3, device=dev), rnd.key(5, dev),
                       torch.zeros((), dtype=torch.int64, device=dev))
    mb = n_local // MB
    rows = []
    if world == 1:
        # union of the 2-rank run's local minibatches (this file is also imported by the
        # single-process reference with world == 1: it then needs the 2-rank permutations)
        perms = local_perms(2)
        half = N // 2
        hmb = half // MB
        for e in range(E):
            for k in range(MB):
                rows.append(torch.cat([r * half + perms[r][e][k * hmb:(k + 1) * hmb]
                                       for r in range(2)]))
    else:
        perms = local_perms(world)[rank]
        for e in range(E):
            for k in range(MB):
                rows.append(perms[e][k * mb:(k + 1) * mb])
    inds = torch.stack(rows).to(dev)
    return env, ts, inds


def run_iterations(env, ts, inds, iters=2):
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.types import LoggingLevel

    from nnx_ppo_amd import _lib, config

    n_local = ts.env_states.done.shape[0]
    level = LoggingLevel.LOSSES | LoggingLevel.ACTOR_EXTRA
    ms = []
    used = set()
    with config.use_compute_dtype("bf16" if BIG else "f32"):
        for it in range(iters):
            if it == iters - 1:
                with _lib.profiler as prof:
                    ts, m = ppo.ppo_step(env, ts, n_local, T, *ARGS, 1.0, level,
                                         minibatch_inds=inds)
                used = sorted({name for name, *_ in prof.records})
            else:
                ts, m = ppo.ppo_step(env, ts, n_local, T, *ARGS, 1.0, level,
                                     minibatch_inds=inds)
            ms.append({k: float(v) for k, v in m.items()})
    norm = ts.networks.layers[0]
    return {"params": ts.optimizer.params.clone(), "adam_m": ts.optimizer.m.clone(),
            "step": int(ts.optimizer.step), "norm_mean": norm.mean.value.clone(),
            "norm_m2": norm.M2.value.clone(), "norm_count": float(norm.counter.value),
            "steps_taken": int(ts.steps_taken), "metrics": ms,
            "obs": ts.env_states.obs.clone(), "used": list(used)}
