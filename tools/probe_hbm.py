"""What this chip sustains for streaming reads / writes / copies at the sizes of the gradient
step's images (50-100 MB), with library kernels inside a captured graph — the practical
ceiling next to which DESIGN.md quotes the trunk and dW kernels' 2.6-3.6 TB/s."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tools._timing import timed  # noqa: E402

dev = torch.device("cuda:0")
for mb in (50, 100, 400):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    out = torch.empty((), device=dev)
    t_r = timed(lambda: torch.sum(x, dim=0, out=out), reps=10)
    t_w = timed(lambda: y.fill_(1.0), reps=10)
    t_c = timed(lambda: y.copy_(x), reps=10)
    gb = n * 4 / 1e9
    print(f"{mb:4d} MB: read {gb / t_r * 1e6:6.0f} GB/s ({t_r:6.1f} us)   write {gb / t_w * 1e6:6.0f} GB/s "
          f"({t_w:6.1f} us)   copy {2 * gb / t_c * 1e6:6.0f} GB/s ({t_c:6.1f} us)", flush=True)
