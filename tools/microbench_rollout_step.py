"""One rollout step of the C2 network (policy step at M = 4096 rows) on the tile kernel
(MIPPO_WS_ROLLOUT=0) or the one-launch weights-stationary form (default), inside a
captured graph of back-to-back calls."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import config  # noqa: E402
from nnx_ppo_amd.networks import factories, policy  # noqa: E402
from nnx_ppo_amd.networks.types import Rngs  # noqa: E402
from tools._timing import timed  # noqa: E402

dev = torch.device("cuda:0")
config.set_compute_dtype("bf16")
for n in (4096, 1024, 8192):
    net = factories.make_mlp_actor_critic(5, 1, [64] * 4, [256] * 2, Rngs(17)).to(dev)
    st = net.initialize_state(n)
    obs = torch.randn(n, 5, device=dev)
    for ws in (False, True):
        policy.WS_POLICY_ROLLOUT = ws
        net(st, obs)
        print(f"n_envs {n:5d}  ws_rollout={ws!s:5s} {timed(lambda: net(st, obs)):7.2f} us", flush=True)
