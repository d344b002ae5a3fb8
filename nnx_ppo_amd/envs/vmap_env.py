"""`VmapEnv`: a single-env, playground-style env lifted to the batched convention.

The reference writes `reset(rng) -> State` / `step(state, action) -> State` for ONE
environment and batches them with `jax.vmap` inside the loop
(`nnx_ppo/algorithms/rollout.py:21,39`, `ppo.py:549`; `docs/reference/batching.rst:70-94`).
This build's loop takes batched envs (leading `n_envs` axis on every leaf,
`algorithms/types.py`); `VmapEnv(env)` is the adapter for code written the reference's
way: it runs `env.reset` / `env.step` under `torch.func.vmap`.

What the single-env code may do is what `jax.vmap` allows, restated for torch:
  * torch operations on its tensors (no `.item()`, no Python branching on tensor values —
    use `torch.where`), returning a `State` (or any dataclass / dict / tuple tree) whose
    leaves are tensors or Python numbers / bools (turned into tensors);
  * key handling through `nnx_ppo_amd.random` (`split`, `fold_in`, `uniform`, `randint`,
    ...): inside the adapter those run as integer torch arithmetic instead of one kernel
    launch each (a vmapped tensor has no device pointer) — same bits;
  * in-place edits of the `info` / `metrics` dicts of the state it was given, as
    `EpisodeWrapper` does in the reference (`episode_wrapper.py:14-22`).
The lifted env is HIP-graph capturable if the single-env code is (no host reads); it is
slower than an env written batched against the kernels (`wrappers/episode_wrapper.py`,
`envs/synthetic.py`): every torch op is its own launch.
"""
from __future__ import annotations

from typing import Any

import torch

from .. import random as rnd
from ..tree import tree_leaves, tree_map


def _is_tensor(x) -> bool:
    return isinstance(x, torch.Tensor)


class VmapEnv:
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):  # observation_size, action_size, ...
        return getattr(self.env, name)

    def _lift(self, fn, *trees):
        """`vmap(fn)(*trees)` for pytrees of this package (dataclass / dict / list nodes):
        tensor leaves are mapped over axis 0, everything else is passed through."""
        leaves = [tree_leaves(t) for t in trees]
        flat = [x for ls in leaves for x in ls if _is_tensor(x)]
        device = flat[0].device
        skeleton: list = [None]

        def call(*tensors):
            it = iter(tensors)
            args = [tree_map(lambda x: next(it) if _is_tensor(x) else x, t) for t in trees]
            out = fn(*args)
            # Python numbers / bools the single-env code returned (reward=1.0, done=False)
            out = tree_map(lambda x: x if _is_tensor(x) else torch.as_tensor(x, device=device),
                           out)
            skeleton[0] = out
            return tuple(tree_leaves(out))

        with rnd.torch_only():
            outs = torch.func.vmap(call, in_dims=0, out_dims=0)(*flat)
        it = iter(outs)
        return tree_map(lambda _: next(it), skeleton[0])

    def reset(self, rng: torch.Tensor) -> Any:
        return self._lift(self.env.reset, rng)

    def step(self, state: Any, action: Any) -> Any:
        return self._lift(self.env.step, state, action)
