"""What the vendor GEMM library reaches on C3's layer shapes (torch.mm -> hipBLASLt /
rocBLAS), next to this package's per-layer kernels: is a library call worth binding for the
plain products of trunks wider than the whole-trunk kernels' class?"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tools._timing import timed  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
for M, K, N in ((61440, 512, 512), (61440, 256, 256), (61440, 17, 512), (30720, 256, 256)):
    x = torch.randn(M, K, device=dev, dtype=bf)
    w = torch.randn(K, N, device=dev, dtype=bf)
    wt = w.t().contiguous()
    dz = torch.randn(M, N, device=dev, dtype=bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    gw = torch.empty(K, N, device=dev, dtype=torch.float32)
    fl = 2.0 * M * K * N
    t = timed(lambda: torch.mm(x, w, out=out), reps=10)
    print(f"fwd  {M}x{K}x{N}: {t:8.1f} us  {fl / t / 1e6:7.1f} TF/s", flush=True)
    t = timed(lambda: torch.mm(dz, wt), reps=10)
    print(f"dX   {M}x{N}x{K}: {t:8.1f} us  {fl / t / 1e6:7.1f} TF/s", flush=True)
    t = timed(lambda: torch.mm(x.t(), dz), reps=10)
    print(f"dW   {K}x{M}x{N}: {t:8.1f} us  {fl / t / 1e6:7.1f} TF/s", flush=True)
