"""us / TF/s of the grouped dW (mi_dense_bwd_dw_grouped_bf16) at BASELINE config 3's shapes:
the critic trunk's and the actor trunk's problems as `dense_chain.backward` hands them over.
MIPPO_GEMM256=0: the 128-row kernel alone."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops  # noqa: E402
from microbench_gemm256 import timed  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    M = 61440
    for name, dims in (("critic 17-512-512-1", [17, 512, 512, 1]),
                       ("actor 17-256x4-12", [17, 256, 256, 256, 256, 12]),
                       ("one 512x512", [512, 512]), ("one 256x256", [256, 256])):
        probs = []
        for k, n in zip(dims[:-1], dims[1:]):
            x = ops.cast_pad_bf16(torch.randn(M, k, device=dev))
            dz = ops.cast_pad_bf16(torch.randn(M, n, device=dev))
            probs.append((x, dz, torch.zeros(k, n, device=dev), torch.zeros(n, device=dev)))
        fl = sum(2.0 * M * k * n for k, n in zip(dims[:-1], dims[1:]))
        t = timed(lambda: ops.dense_bwd_dw_grouped_bf16(probs, accumulate=True), iters=10)
        print(json.dumps({"problems": name, "M": M, "us": round(t, 1),
                          "TFs": round(fl / t / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
