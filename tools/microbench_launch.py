"""Launch-cost microbenchmark on the target box: per-kernel cost of trivial
kernels launched eagerly through the C ABI vs replayed from a captured HIP
graph (serial chain and two parallel branches)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
st = ops.make_rng_state(1, dev)
st2 = ops.make_rng_state(2, dev)
N = 2000


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def eager():
    for _ in range(N):
        ops.rng_advance(st, 1)


print("eager    : %.2f us / kernel" % (timeit(eager) / N * 1e6))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eager()
print("graph    : %.2f us / kernel (serial chain of %d)" % (timeit(g.replay) / N * 1e6, N))

g2 = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.graph(g2):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(N // 2):
            ops.rng_advance(st2, 1)
    for _ in range(N // 2):
        ops.rng_advance(st, 1)
    main.wait_stream(side)
print("graph x2 : %.2f us / kernel (two parallel branches of %d)" % (timeit(g2.replay) / N * 1e6, N // 2))

x = torch.zeros(4096, device=dev)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    for _ in range(N):
        x.add_(1.0)
print("graph torch add_: %.2f us / kernel" % (timeit(g3.replay) / N * 1e6))
