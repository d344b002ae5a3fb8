"""GAE + loss as two launches against the one-launch form, in isolation (a captured graph
of back-to-back calls, as the iteration runs them)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops  # noqa: E402
from tools._timing import timed  # noqa: E402

dev = torch.device("cuda:0")
T, N = 30, 1024
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
r, v, lv = rn(T, N), rn(T, N), rn(N)
done = torch.rand(T, N, device=dev, generator=g) < 0.1
trunc = done & (torch.rand(T, N, device=dev, generator=g) < 0.5)
llo = rn(T, N) - 1
lln = llo + 0.1 * rn(T, N)
reg = rn(T, N)


def two(norm=True):
    if norm:
        adv, st = ops.gae(r, v, lv, done, trunc, 0.99, 0.95, with_stats=True)
    else:
        adv, st = ops.gae(r, v, lv, done, trunc, 0.99, 0.95), None
    ops.ppo_loss(lln.reshape(-1), llo.reshape(-1), adv.reshape(-1), v.reshape(-1), reg.reshape(-1),
                 st, 0.2, 1.0)


def one(norm=True):
    ops.gae_ppo_loss(r, v, lv, done, trunc, lln, llo, reg, 0.99, 0.95, norm, 0.2, 1.0)


for name, fn in (("two launches, normalised", lambda: two(True)),
                 ("two launches, raw", lambda: two(False)),
                 ("one launch, normalised", lambda: one(True)),
                 ("one launch, raw (no exchange)", lambda: one(False))):
    print(f"{name:34s} {timed(fn):7.2f} us", flush=True)
