"""Phase timeline of the weights-stationary trunk kernel (a -DMIPPO_TRACE build):

    MIPPO_LIB=ab/libmippo_trace.so python tools/trace_ws.py [critic|actor] [M]

Thread 0 of every workgroup stamps the shader clock at each phase boundary; prints the
mean cycles between consecutive stamps over the workgroups."""
import ctypes
import math
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops  # noqa: E402

EV, WG = 32, 512
name = sys.argv[1] if len(sys.argv) > 1 else "critic"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 122880
dims = {"critic": [5, 256, 256, 1], "actor": [5, 64, 64, 64, 64, 2]}[name]
dev = torch.device("cuda:0")
BF = torch.bfloat16
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
x = torch.randn(M, dims[0], device=dev)
cd = ctypes.CDLL(os.environ["MIPPO_LIB"])
cd.mi_debug_ws_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
for _ in range(3):
    ops.mlp_ws_fwd_bf16(x, ffs, bs, dims, acts, train=True)
torch.cuda.synchronize()
buf = np.zeros(WG * EV, dtype=np.uint64)
cd.mi_debug_ws_trace(buf.ctypes.data, buf.size)          # clear
ops.mlp_ws_fwd_bf16(x, ffs, bs, dims, acts, train=True)
torch.cuda.synchronize()
assert cd.mi_debug_ws_trace(buf.ctypes.data, buf.size) == 0
tr = buf.reshape(WG, EV).astype(np.int64)
live = tr[:, 0] != 0
tr = tr[live]
d = np.diff(tr, axis=1)
d[tr[:, 1:] == 0] = 0
print(f"{name} M={M}: {int(live.sum())} workgroups; mean cycles between stamps:")
print(" ", [int(v) for v in d.mean(0)])
print("  start spread (cycles):", int(tr[:, 0].max() - tr[:, 0].min()))
