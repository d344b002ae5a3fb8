"""The tail of a C2 gradient step in isolation: grouped dW (8 layers, M = 30720) followed
by the optimiser, with the slab reduction in its own launch against folded into the Adam
launch (a captured graph of back-to-back calls, as the iteration runs them)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops, optim  # noqa: E402
from nnx_ppo_amd.networks import factories  # noqa: E402
from nnx_ppo_amd.networks.types import Rngs  # noqa: E402
from tools._timing import timed  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 30720
net = factories.make_mlp_actor_critic(5, 1, [64] * 4, [256] * 2, Rngs(17)).to(dev)
opt = optim.Optimizer(net, 1e-4)
named = dict(net.named_parameters())
layers = [(p, named.get(k[:-len("kernel")] + "bias")) for k, p in named.items()
          if k.endswith("kernel") and len(p.shape) == 2]
rng = np.random.default_rng(0)
problems = []
for w, b in layers:
    K, N = w.shape
    x = ops.cast_pad_bf16(torch.as_tensor(rng.normal(size=(M, K)).astype(np.float32)).to(dev))
    dz = ops.cast_pad_bf16(torch.as_tensor(rng.normal(size=(M, N)).astype(np.float32)).to(dev) * 1e-3)
    problems.append((x, dz, w.grad, b.grad))


def step(defer):
    opt.begin(defer_dw=defer)
    ops.dense_bwd_dw_grouped_bf16(problems, accumulate=True)
    opt.update()


def dw_only():
    ops.dense_bwd_dw_grouped_bf16(problems, accumulate=True)


def adam_only():
    opt.begin()
    opt.update()


print(f"arena {opt.n} floats, M = {M}")
for name, fn in (("dW + reduce", dw_only), ("adam alone", adam_only),
                 ("dW + reduce + adam", lambda: step(False)),
                 ("dW + adam(slabs)", lambda: step(True))):
    print(f"{name:24s} {timed(fn):7.2f} us", flush=True)
