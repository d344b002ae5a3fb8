"""Synthetic torch environments with the playground API (batched convention,
see `nnx_ppo_amd/algorithms/types.py`).  They restate the reference's
closed-form test envs (`nnx_ppo/test_dummies/`) — its "fake backends" — and
provide the Cartpole-/Cheetah-shaped workloads the benchmark is quoted on
(real mujoco_playground envs are JAX programs and cannot run here)."""
from .synthetic import (DummyCounterEnv, MockEnv, MoveToCenterEnv, TwoArmEnv,  # noqa: F401
                        cartpole_shaped, cheetah_shaped)
from .vmap_env import VmapEnv  # noqa: F401
