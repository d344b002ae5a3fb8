"""Phase timeline of the weights-stationary BACKWARD kernel (a -DMIPPO_TRACE build):

    MIPPO_LIB=ab/libmippo_trace.so python tools/trace_ws_bwd.py [critic|actor] [M]

Stamps (thread 0 of every workgroup): 0 start, 1 weights requested, then per row tile:
+0 tile start, +1 head gradient staged / aux requested, +2 barrier, +3 head multiplied,
+4 epilogue (waits for the aux loads), +5 barrier, +6 copy-out issued, then per hidden
layer: multiplied, epilogue, barrier, copy-out issued; tile end."""
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
M = int(sys.argv[2]) if len(sys.argv) > 2 else 30720
dims = {"critic": [5, 256, 256, 1], "actor": [5, 64, 64, 64, 64, 2]}[name]
dev = torch.device("cuda:0")
BF = torch.bfloat16
L = len(dims) - 1
acts = [ops.ACT_RELU] * (L - 1) + [ops.ACT_NONE]
ffs, fbs, bs = [], [], []
for l in range(L):
    K, N = dims[l], dims[l + 1]
    w = torch.randn(K, N, device=dev) / math.sqrt(K)
    w_bf = torch.zeros(K, ops.pad8(N), dtype=BF, device=dev)
    wt_bf = torch.zeros(N, ops.pad8(K), dtype=BF, device=dev)
    nf, nb = ops.frag_sizes(K, N)
    ff, fb = torch.zeros(nf, dtype=BF, device=dev), torch.zeros(nb, dtype=BF, device=dev)
    ops.weights_to_bf16_multi([w], [w_bf], [wt_bf], [ff], [fb])
    ffs.append(ff)
    fbs.append(fb)
    bs.append(torch.zeros(N, device=dev))
x = torch.randn(M, dims[0], device=dev)
_, saved = ops.mlp_fwd_bf16(x, ffs, bs, dims, acts, train=True)
auxs = [sv[1] for sv in saved]
g = torch.randn(M, dims[-1], device=dev) / M
cd = ctypes.CDLL(os.environ["MIPPO_LIB"])
cd.mi_debug_ws_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
run = lambda: ops.mlp_ws_bwd_dx_bf16(g, fbs, dims, acts, auxs)
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = np.zeros(WG * EV, dtype=np.uint64)
cd.mi_debug_ws_trace(buf.ctypes.data, buf.size)          # clear
run()
torch.cuda.synchronize()
assert cd.mi_debug_ws_trace(buf.ctypes.data, buf.size) == 0
tr = buf.reshape(WG, EV).astype(np.int64)
live = tr[:, 0] != 0
tr = tr[live]
d = np.diff(tr, axis=1)
d[tr[:, 1:] == 0] = 0
print(f"{name} backward M={M}: {int(live.sum())} workgroups; mean cycles between stamps:")
print(" ", [int(v) for v in d.mean(0)])
last = np.array([row[row != 0][-1] - row[0] for row in tr])
print("  workgroup life (cycles): mean", int(last.mean()), "max", int(last.max()),
      " start spread:", int(tr[:, 0].max() - tr[:, 0].min()))
