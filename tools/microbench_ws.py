"""Weights-stationary trunk forward against the per-tile trunk kernel (training mode,
every image kept), C2's two trunks at the replay size."""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops  # noqa: E402
from tools._timing import timed  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
for name, dims in (("critic 5-256-256-1", [5, 256, 256, 1]), ("actor 5-64x4-2", [5, 64, 64, 64, 64, 2])):
    L = len(dims) - 1
    acts = [ops.ACT_RELU] * (L - 1) + [ops.ACT_NONE]
    ffs, bs = [], []
    for l in range(L):
        K, N = dims[l], dims[l + 1]
        w = torch.randn(K, N, device=dev) / math.sqrt(K)
        w_bf = torch.zeros(K, ops.pad8(N), dtype=BF, device=dev)
        wt_bf = torch.zeros(N, ops.pad8(K), dtype=BF, device=dev)
        nf, nb = ops.frag_sizes(K, N)
        ff, fb = torch.zeros(nf, dtype=BF, device=dev), torch.zeros(nb, dtype=BF, device=dev)
        ops.weights_to_bf16_multi([w], [w_bf], [wt_bf], [ff], [fb])
        ffs.append(ff)
        bs.append(torch.zeros(N, device=dev))
    for M in (30720, 31744, 8192, 122880):
        x = torch.randn(M, dims[0], device=dev)
        a = timed(lambda: ops.mlp_fwd_bf16(x, ffs, bs, dims, acts, train=True))
        b = timed(lambda: ops.mlp_ws_fwd_bf16(x, ffs, bs, dims, acts, train=True))
        kept = sum(2 * M * ops.pad8(d) for d in dims[:-1])
        print(f"{name:20s} M={M:6d}  tile kernel {a:7.2f} us   weights-stationary {b:7.2f} us   "
              f"({(4 * M * dims[0] + kept + 4 * M * dims[-1]) / b / 1e3:6.0f} GB/s)", flush=True)
