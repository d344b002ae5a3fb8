"""Per-C-ABI-entry device time of ONE eager iteration of a BASELINE config (HIP events around
every call, queued behind a spin kernel so that host gaps do not count) — the quick view of
where a config's time goes.   python tools/profile_config.py C3|C4"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from nnx_ppo_amd import _lib, config as mi_config  # noqa: E402
from nnx_ppo_amd.algorithms import ppo  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    mi_config.set_compute_dtype("bf16")
    dev = torch.device("cuda", 0)
    n = bench.OTHER_CONFIGS[name]["n_envs"]
    env, net, ts = bench.build_config(name, n, dev)
    step = lambda st: ppo.ppo_step(env, st, n, bench.T, 0.95, 0.99, 0.2, True, False,
                                   bench.N_EPOCHS, bench.N_MB)
    for _ in range(2):
        ts, _ = step(ts)
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2.0e9 * 0.25))
    with _lib.profiler as prof:
        ts, _ = step(ts)
    summ = prof.summary()
    tot = sum(d["ms"] for d in summ.values())
    print(f"{name}: {tot:.3f} ms of C-ABI device time in one iteration (torch ops not included)")
    for k, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:24]:
        print(f"  {k:44s} {d['calls']:5d} x {1e3 * d['ms'] / d['calls']:8.1f} us = {d['ms']:8.3f} ms")


if __name__ == "__main__":
    main()
