"""Twenty training-size policy forwards (C2 network) — the target of `rocprofv3 --stats`."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import config  # noqa: E402
from nnx_ppo_amd.networks import factories  # noqa: E402
from nnx_ppo_amd.networks.types import PPONetworkOutput, Rngs  # noqa: E402
from nnx_ppo_amd.optim import Optimizer  # noqa: E402

dev = torch.device("cuda:0")
config.set_compute_dtype("bf16")
T, B = 30, 1024
net = factories.make_mlp_actor_critic(5, 1, [64] * 4, [256] * 2, Rngs(17))
net.to(dev)
opt = Optimizer(net, 1e-4, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
obs, last = rn(T, B, 5), rn(B, 5)
done = torch.zeros(T, B, dtype=torch.bool, device=dev)
extras = [obs, {"action": [None] * 5 + [rn(T, B, 1)], "value": [None] * 3}]
st = net.initialize_state(B)
g_out = PPONetworkOutput(None, rn(T, B), rn(T, B))
for _ in range(20):
    r = net.replay_with_bootstrap(st, obs, done, extras, last)
    opt.begin()
    net.replay_backward(r[0], g_out, 1.0 / (T * B))
torch.cuda.synchronize()
