"""Per-workgroup phase timeline of the policy kernels (a -DMIPPO_TRACE build of mlp_bf16.hip):

    MIPPO_LIB=ab/libmippo_trace.so python tools/trace_policy.py [T] [minibatch]

Thread 0 of every workgroup stamps the shader clock at each phase boundary; this prints,
per trunk (blockIdx.y), the mean cycles between consecutive stamps and the spread of
workgroup start / end times on the 100 MHz wall clock.
"""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import _lib, config as mi_config  # noqa: E402
from nnx_ppo_amd.networks import factories  # noqa: E402
from nnx_ppo_amd.networks.types import Rngs  # noqa: E402

TR_EV, TR_WG = 64, 2048
T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
mi_config.set_compute_dtype("bf16")
net = factories.make_mlp_actor_critic(5, 1, [64] * 4, [256] * 2, Rngs(3))
net.to(dev)
from nnx_ppo_amd.optim import Optimizer  # noqa: E402

opt = Optimizer(net, 1e-3, None, None, device=dev)
cd = ctypes.CDLL(os.environ["MIPPO_LIB"])
cd.mi_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64]


def dump(tag, rows_per_wg, rows_total):
    torch.cuda.synchronize()
    buf = np.zeros(TR_WG * TR_EV, dtype=np.uint64)
    rc = cd.mi_debug_trace(buf.ctypes.data, buf.size)
    assert rc == 0, rc
    tr = buf.reshape(TR_WG, TR_EV).astype(np.int64)
    live = tr[:, 0] != 0
    print(f"== {tag}: {int(live.sum())} workgroups traced")
    cd.mi_debug_trace_clear()
    t0 = tr[live, TR_EV - 2].min()
    for name, sel in (("all", live),):
        st = (tr[sel, TR_EV - 2] - t0) / 100.0
        en = (tr[sel, TR_EV - 1] - t0) / 100.0
        print(f"  wall-clock us: first start {st.min():.1f}, last start {st.max():.1f}, "
              f"first end {en.min():.1f}, last end {en.max():.1f}, mean life {np.mean(en - st):.1f}")
        hist, edges = np.histogram(st, bins=10)
        print("  start histogram (us):", [f"{e:.0f}:{h}" for h, e in zip(hist, edges)])
    # blockIdx.y = 0: action trunk, 1: value trunk (workgroup index = y * gridDim.x + x)
    X = -(-rows_total // rows_per_wg)
    nev = (tr[:, :TR_EV - 2] != 0).sum(1)
    for y, name in ((0, "action"), (1, "value")):
        sel = live.copy()
        sel[:y * X] = False
        sel[(y + 1) * X:] = False
        if not sel.any():
            continue
        n = int(nev[sel].min())
        d = np.diff(tr[sel, :n], axis=1)
        life = (tr[sel, TR_EV - 1] - tr[sel, TR_EV - 2]) / 100.0
        print(f"  {name} trunk ({n} stamps): {int(sel.sum())} wgs, life {life.mean():.1f} us "
              f"(min {life.min():.1f}, max {life.max():.1f}); total cycles {d.sum(1).mean():.0f}")
        print("    mean cycles per phase:", [int(v) for v in d.mean(0)])


state = net.initialize_state(B)
x = torch.randn(T, B, 5, device=dev)
done = torch.zeros(T, B, dtype=torch.bool, device=dev)
# rollout form first (also gives the extras the replay needs)
outs = []
s = state
for t in range(T):
    o = net(s, x[t])
    s = o.next_state
    outs.append(o.rollout_extras)
NR = 4096
sr = net.initialize_state(NR)
xr = torch.randn(NR, 5, device=dev)
for _ in range(3):
    net(sr, xr)
cd.mi_debug_trace_clear()
torch.cuda.synchronize()
net(sr, xr)
dump("policy_kernel<1> rollout step (M = %d)" % NR, 16, NR)
from nnx_ppo_amd.tree import tree_map  # noqa: E402

extras = tree_map(lambda *xs: torch.stack(xs, 0), outs[0], *outs[1:])
last = torch.randn(B, 5, device=dev)
for _ in range(3):
    r = net.replay_with_bootstrap(state, x, done, extras, last)
dump("policy_kernel<4> replay (M = %d + %d)" % (T * B, B), 64, T * B + B)
ctx, out, reg, fs, lv = r
from nnx_ppo_amd.networks.types import PPONetworkOutput  # noqa: E402

g = PPONetworkOutput(None, torch.randn(T, B, device=dev), torch.randn(T, B, device=dev))
opt.begin()
net.replay_backward(ctx, g, 1.0 / (T * B))
dump("policy_bwd_kernel<4> (M = %d)" % (T * B), 64, T * B)

# ---- dW kernel of the same backward (gemm_bf16.hip, -DMIPPO_TRACE) ---------------------
cd.mi_debug_trace_dw.argtypes = [ctypes.c_void_p, ctypes.c_int64]
cd.mi_debug_trace_dw_clear()
torch.cuda.synchronize()
opt.begin()
net.replay_backward(ctx, g, 1.0 / (T * B))
torch.cuda.synchronize()
buf = np.zeros(2048 * 12, dtype=np.uint64)
assert cd.mi_debug_trace_dw(buf.ctypes.data, buf.size) == 0
tr = buf.reshape(2048, 12).astype(np.int64)
live = tr[:, 0] != 0
print(f"== tn_gemm_dw_all_kernel: {int(live.sum())} working workgroups traced")
w0 = tr[live, 4].min()
st, en = (tr[live, 4] - w0) / 100.0, (tr[live, 5] - w0) / 100.0
print(f"  wall-clock us: last start {st.max():.1f}, first end {en.min():.1f}, last end {en.max():.1f}")
for tag in sorted(set(tr[live, 6])):
    sel = live & (tr[:, 6] == tag)
    pro = (tr[sel, 1] - tr[sel, 0]).mean()
    loop = (tr[sel, 2] - tr[sel, 1]).mean()
    epi = (tr[sel, 3] - tr[sel, 2]).mean()
    its = tr[sel, 7].mean()
    life = ((tr[sel, 5] - tr[sel, 4]) / 100.0)
    print(f"  class WM*100+TN={int(tag)}: {int(sel.sum())} wgs, life {life.mean():.1f} us "
          f"(max {life.max():.1f}); cycles: prologue {pro:.0f}, loop {loop:.0f} "
          f"({its:.1f} tiles, {loop / max(its, 1):.0f} per tile), epilogue {epi:.0f}")
    half = np.maximum((tr[sel, 7] + 1) // 2, 1)[:, None]
    ph = (tr[sel, 8:12] / half).mean(0)
    print(f"    per even tile: multiply {ph[0]:.0f}, store to LDS (+ wait for its loads) "
          f"{ph[1]:.0f}, load issue {ph[2]:.0f}, barrier {ph[3]:.0f}")

