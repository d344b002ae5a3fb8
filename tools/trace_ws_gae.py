"""Phase timeline of the backward launch with the GAE inside (a -DMIPPO_TRACE build):

    MIPPO_LIB=ab/libmippo_trace.so python tools/trace_ws_gae.py [T] [B]

Workgroups 0 .. n_value-1 are the value trunk.  Stamps (thread 0): 0 start; action trunk: 1
slot table, then one per chunk of 8 env groups scanned, then all scans done; value trunk: 1, 2
own tiles scanned, 3 all done; then `weights requested` and the per-tile stamps of
tools/trace_ws_bwd.py."""
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
T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
M = T * B
dev = torch.device("cuda:0")
BF = torch.bfloat16


def images(dims):
    L = len(dims) - 1
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
    return ffs, fbs, bs, [ops.ACT_RELU] * (L - 1) + [ops.ACT_NONE]


a_dims, c_dims = [5, 64, 64, 64, 64, 2], [5, 256, 256, 1]
a_ff, a_fb, a_b, a_acts = images(a_dims)
c_ff, c_fb, c_b, c_acts = images(c_dims)
obs, extras, tail = torch.randn(M, 5, device=dev), torch.randn(M, 1, device=dev), \
    torch.randn(B, 5, device=dev)
rng_state = ops.make_rng_state(99, dev, 3)
kw = dict(min_std=0.1, std_scale=1.0, entropy_weight=1e-2)
rw = ops.policy_fwd_bf16(obs, None, (a_ff, a_b, a_dims, a_acts), (c_ff, c_b, c_dims, c_acts),
                         rng_state, 2, deterministic=False, extras=extras, train=True,
                         want_stats=False, ws=True, value_tail=tail, **kw)
masks = (rw["actor_masks"], rw["critic_masks"])
actor = (a_fb, a_dims, a_acts, [sv[1] for sv in rw["actor_saved"]])
critic = (c_fb, c_dims, c_acts, [sv[1] for sv in rw["critic_saved"]])
values = rw["value"].view(T, B).contiguous()
last_value = rw["value_tail_out"].view(B).contiguous()
ll_new = rw["log_likelihood"].view(T, B).contiguous()
ll_old = (ll_new + 0.3 * torch.randn(T, B, device=dev)).contiguous()
reg = rw["reg"].view(T, B).contiguous()
rewards = torch.randn(T, B, device=dev)
done = torch.rand(T, B, device=dev) < 0.15
trunc = (torch.rand(T, B, device=dev) < 0.05) & done
run = lambda: ops.policy_bwd_gae_bf16(
    rw["mean_and_std"], extras, rng_state, 2, 1.0 / M, actor, critic, masks, rewards, values,
    last_value, done, trunc, ll_new, ll_old, reg, 0.99, 0.95, True, 0.2, 0.5, **kw)
cd = ctypes.CDLL(os.environ["MIPPO_LIB"])
cd.mi_debug_ws_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = np.zeros(WG * EV, dtype=np.uint64)
cd.mi_debug_ws_trace(buf.ctypes.data, buf.size)          # clear
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
print("launch (events, incl. host enqueue):", round(1e3 * e0.elapsed_time(e1), 1), "us")
assert cd.mi_debug_ws_trace(buf.ctypes.data, buf.size) == 0
tr = buf.reshape(WG, EV).astype(np.int64)
live = tr[:, 0] != 0
n_live = int(live.sum())
nv = n_live // 2
for name, sl in (("value", slice(0, nv)), ("action", slice(nv, n_live))):
    t = tr[:n_live][sl]
    d = np.diff(t, axis=1)
    d[t[:, 1:] == 0] = 0
    print(f"{name} trunk, {t.shape[0]} workgroups; mean cycles between stamps:")
    print(" ", [int(v) for v in d.mean(0)])
    last = np.array([row[row != 0][-1] - row[0] for row in t])
    print("  workgroup life: mean", int(last.mean()), "max", int(last.max()), " start spread:",
          int(t[:, 0].max() - t[:, 0].min()))
