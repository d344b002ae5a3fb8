"""One trunk configuration, a few launches — the target of PMC passes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
name = sys.argv[1] if len(sys.argv) > 1 else "critic"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 30720
dims = {"actor": [5, 64, 64, 64, 64, 2], "critic": [5, 256, 256, 1]}[name]
L = len(dims) - 1
acts = [ops.ACT_RELU] * (L - 1) + [ops.ACT_NONE]
ffs, fbs, bs = [], [], []
for l in range(L):
    K, N = dims[l], dims[l + 1]
    w = torch.randn(K, N, device=dev) / K ** 0.5
    w_bf = torch.zeros(K, ops.pad8(N), dtype=BF, device=dev)
    wt_bf = torch.zeros(N, ops.pad8(K), dtype=BF, device=dev)
    nf, nb = ops.frag_sizes(K, N)
    ff, fb = torch.zeros(nf, dtype=BF, device=dev), torch.zeros(nb, dtype=BF, device=dev)
    ops.weights_to_bf16_multi([w], [w_bf], [wt_bf], [ff], [fb])
    ffs.append(ff)
    fbs.append(fb)
    bs.append(torch.zeros(N, device=dev))
x = torch.randn(M, dims[0], device=dev)
g = torch.randn(M, dims[-1], device=dev)
for _ in range(5):
    _, saved = ops.mlp_fwd_bf16(x, ffs, bs, dims, acts, train=True)
    ops.mlp_bwd_dx_bf16(g, None, ops.ACT_NONE, fbs, dims, acts, [sv[1] for sv in saved], False)
torch.cuda.synchronize()
