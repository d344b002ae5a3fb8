"""mi_key_permutations: rank kernel vs the one-workgroup bitonic network (set
MIPPO_PERM_BITONIC=1 for the latter), 4 permutations of n."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops, random as rnd  # noqa: E402
from tools._timing import timed  # noqa: E402

dev = torch.device("cuda:0")
k = rnd.key(5, dev)
for n in (1024, 4096, 8192):
    print(n, f"{timed(lambda: ops.key_permutations(k, 4, n)):.2f} us", flush=True)
