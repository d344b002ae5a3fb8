"""What the GPU box's host offers the CPU baseline: affinity, cgroup quota, and the CPU
oracle's iteration time at a few thread counts (one timed iteration each)."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import bench  # noqa: E402

print("usable:", bench._usable_cores(), "cpu_count", os.cpu_count(), flush=True)
for p in ("/sys/fs/cgroup/cpu.max", "/proc/loadavg"):
    try:
        print(p, Path(p).read_text().strip(), flush=True)
    except OSError as e:
        print(p, e)
from nnx_ppo_amd import random as keys  # noqa: E402
from nnx_ppo_amd.envs import cartpole_shaped  # noqa: E402
from nnx_ppo_amd.networks import factories  # noqa: E402
from nnx_ppo_amd.networks.types import Rngs  # noqa: E402
from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper  # noqa: E402
from oracle import networks as on  # noqa: E402
from oracle import ppo as op  # noqa: E402

aff = len(os.sched_getaffinity(0))
for threads in [t for t in (8, 16, 32, 64, aff) if t <= aff]:
    torch.set_num_threads(threads)
    net = factories.make_mlp_actor_critic(5, 1, [64] * 4, [256] * 2, Rngs(17))
    onet = on.from_product(net, torch.float32)
    env = EpisodeWrapper(cartpole_shaped(max_steps=1000), 1000)
    ts = op.new_training_state(env, onet, 4096, 17, keys)
    t0 = time.perf_counter()
    ts, _ = op.ppo_step(env, ts, 4096, 30, 0.95, 0.99, 0.2, True, 4, 4, keys)
    t1 = time.perf_counter()
    ts, _ = op.ppo_step(env, ts, 4096, 30, 0.95, 0.99, 0.2, True, 4, 4, keys)
    t2 = time.perf_counter()
    print(f"threads {threads}: first {t1 - t0:.2f} s, second {t2 - t1:.2f} s", flush=True)
