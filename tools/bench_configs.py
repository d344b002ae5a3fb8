"""env-steps/s of the other BASELINE configs and of a large-N sweep (one GPU, one HIP
graph per iteration) — the numbers DESIGN.md quotes beside bench.py's headline C2 line.

    python tools/bench_configs.py c2:4096 c2:16384 c2:65536 c3 c4
"""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

T, N_EPOCHS, N_MB, SEED = 30, 4, 4, 17


def build(name, n_envs, device):
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import cartpole_shaped, cheetah_shaped
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.adapter import PPOAdapter
    from nnx_ppo_amd.networks.containers import Sequential
    from nnx_ppo_amd.networks.normalizer import Normalizer
    from nnx_ppo_amd.networks.sampling_layers import NormalTanhSampler
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.networks.utils import Flattener
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    rngs = Rngs(SEED)
    if name == "c2":
        env = EpisodeWrapper(cartpole_shaped(max_steps=1000), 1000)
        net = factories.make_mlp_actor_critic(5, 1, [64] * 4, [256] * 2, rngs)
        desc = "CartpoleBalance-shaped, MLP 4x64 / 2x256"
    elif name == "c3":
        env = EpisodeWrapper(cheetah_shaped(max_steps=1000), 1000)
        # a dict obs_size: Sequential([Normalizer(tree), Flattener(), PPOAdapter(...)])
        net = factories.make_mlp_actor_critic({"position": 8, "velocity": 9}, 6, [256] * 4,
                                              [512] * 2, rngs)
        desc = "CheetahRun-shaped dict obs {position 8, velocity 9}, MLP 4x256 / 2x512"
    elif name in ("c4", "c4r"):
        # c4r: the reset-heavy env of SURVEY 8(d) (an episode ends every 5 steps, so the
        # carry reset inside the BPTT kernels is exercised on every replay)
        inner = 5 if name == "c4r" else 1000
        env = EpisodeWrapper(cartpole_shaped(max_steps=inner), 1000)
        net = factories.make_gru_actor_critic(5, 1, 64, [256, 256], rngs)
        desc = ("CartpoleBalance-shaped" + (", max_steps=5 (reset-heavy)" if name == "c4r" else "")
                + ", actor Dense-GRU(64)-Dense / critic 2x256")
    else:
        raise SystemExit(f"unknown config {name}")
    ts = ppo.new_training_state(env, net, n_envs, SEED, 1e-4, device=device)
    return env, ts, desc


def main():
    from nnx_ppo_amd import config as mi_config
    from nnx_ppo_amd.algorithms.graph import GraphedPPOStep

    mi_config.set_compute_dtype("bf16")
    device = torch.device("cuda", 0)
    default_n = {"c2": 4096, "c3": 8192, "c4": 4096, "c4r": 4096}
    for spec in sys.argv[1:] or ["c2", "c3", "c4"]:
        name, _, n = spec.partition(":")
        n_envs = int(n) if n else default_n[name]
        env, ts, desc = build(name, n_envs, device)
        step = GraphedPPOStep(env, ts, n_envs, T, 0.95, 0.99, 0.2, True, False, N_EPOCHS, N_MB,
                              warmup=2)
        for _ in range(3):
            ts_, m = step()
            _ = int(ts_.steps_taken)
        torch.cuda.synchronize()
        iters = 20 if n_envs <= 16384 else 8
        t0 = time.perf_counter()
        for _ in range(iters):
            ts_, m = step()
            _ = int(ts_.steps_taken)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"config": name, "workload": desc, "n_envs": n_envs, "rollout_length": T,
                          "env_steps_per_s": round(n_envs * T * iters / dt, 1),
                          "ms_per_iter": round(dt / iters * 1e3, 3), "iters": iters,
                          "dtype": "bf16", "launch": "hip-graph",
                          "losses_actor_mean": float(m["losses/actor/mean"])}), flush=True)
        del step, ts, env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
