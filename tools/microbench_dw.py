"""Grouped dW launches of the C2 / C3 trunks in isolation (one stream)."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


M = int(os.environ.get("M", 30720))
for name, dims in [("actor", [5, 64, 64, 64, 64, 2]), ("critic", [5, 256, 256, 1]),
                   ("critic_L2_only", [256, 256]), ("c3actor", [17, 256, 256, 256, 256, 12]),
                   ("c3critic", [17, 512, 512, 1])]:
    L = len(dims) - 1
    probs = []
    for l in range(L):
        K, N = dims[l], dims[l + 1]
        x = torch.randn(M, ops.pad8(K), device=dev).to(BF)
        dz = torch.randn(M, ops.pad8(N), device=dev).to(BF)
        probs.append((x, dz, torch.zeros(K, N, device=dev), torch.zeros(N, device=dev)))
    # operand widths are carried by the grad shapes
    us = timed(lambda: ops.dense_bwd_dw_grouped_bf16(probs, accumulate=True))
    byt = sum((p[0].numel() + p[1].numel()) * 2 for p in probs)
    print({"chain": name, "M": M, "blocks": os.environ.get("MIPPO_DW_BLOCKS", "512"),
           "us": round(us, 1), "operand_GBps": round(byt / us / 1e3, 1)}, flush=True)
