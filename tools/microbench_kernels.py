"""Kernel-level roofline sweep on the target box (not part of bench.py's contract).

Times the HBM-bound kernels (GAE scan, loss, Adam, normaliser) and the bf16 MFMA
kernels at the workload's size and at sizes large enough to leave the
launch-bound regime, with HIP events around `reps` back-to-back launches, and
prints achieved GB/s (algorithmic bytes) or TFLOP/s against the
MI355X peaks (HBM 8 TB/s spec / 6.3 TB/s measured copy; bf16 MFMA 2.5 PF/s)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3  # seconds per launch


out = []


def hbm(name, nbytes, sec):
    out.append({"kernel": name, "bytes": nbytes, "us": round(sec * 1e6, 2),
                "GB/s": round(nbytes / sec / 1e9, 1), "frac_of_8TB/s": round(nbytes / sec / 8e12, 4)})


def mfma(name, flops, sec):
    out.append({"kernel": name, "flop": flops, "us": round(sec * 1e6, 2),
                "TFLOP/s": round(flops / sec / 1e12, 1),
                "frac_of_2.5PF/s": round(flops / sec / 2.5e15, 4)})


for T, N in [(30, 1024), (30, 4096), (30, 1 << 18), (30, 1 << 22)]:
    r = torch.randn(T, N, device=dev)
    v = torch.randn(T, N, device=dev)
    lv = torch.randn(N, device=dev)
    d = torch.rand(T, N, device=dev) < 0.05
    adv = torch.empty_like(r)
    hbm(f"mi_gae_f32[{T},{N}]", T * N * 14 + 4 * N,
        timed(lambda: ops.gae(r, v, lv, d, d, 0.99, 0.95, out=adv)))
    n = T * N
    if n <= (1 << 25):
        ll = torch.randn(n, device=dev)
        st = ops.adv_stats(adv.view(-1))
        hbm(f"mi_ppo_loss_f32[{n}]", n * 28,
            timed(lambda: ops.ppo_loss(ll, ll, adv.view(-1), v.view(-1), ll, st, 0.2, 1.0)))

for n in [80579, 1 << 24, 1 << 27]:
    p = torch.randn(n, device=dev)
    g = torch.randn(n, device=dev)
    m = torch.zeros(n, device=dev)
    vv = torch.zeros(n, device=dev)
    step = torch.ones(1, dtype=torch.int64, device=dev)
    hbm(f"mi_adam_step_f32[{n}]", n * 28, timed(lambda: ops.adam_step(p, g, m, vv, step, lr=1e-4)))

for M, F in [(122880, 5), (1 << 24, 17)]:
    x = torch.randn(M, F, device=dev)
    mean = torch.zeros(F, device=dev)
    m2 = torch.ones(F, device=dev)
    c = torch.ones(1, device=dev)
    o = torch.empty_like(x)
    hbm(f"mi_normalize_fwd_f32[{M},{F}]", M * F * 8,
        timed(lambda: ops.normalize_fwd(x, mean, m2, c, 1e-6, out=o)))
    hbm(f"mi_welford_batch_stats_f32[{M},{F}]", M * F * 4,
        timed(lambda: ops.welford_batch_stats(x, F)))

for M, K, N in [(30720, 64, 64), (30720, 256, 256), (30720, 512, 512), (245760, 256, 256),
                (245760, 512, 512)]:
    x = ops.cast_pad_bf16(torch.randn(M, K, device=dev))
    w = torch.randn(K, N, device=dev) / K ** 0.5
    w_bf = torch.zeros(K, ops.pad8(N), dtype=torch.bfloat16, device=dev)
    wt_bf = torch.zeros(N, ops.pad8(K), dtype=torch.bfloat16, device=dev)
    ops.weights_to_bf16(w, w_bf, wt_bf)
    b = torch.zeros(N, device=dev)
    dz = ops.cast_pad_bf16(torch.randn(M, N, device=dev))
    gw = torch.zeros(K, N, device=dev)
    fl = 2.0 * M * K * N
    mfma(f"mi_dense_fwd_bf16[{M},{K},{N}]", fl,
         timed(lambda: ops.dense_fwd_bf16(x, wt_bf, b, K, N, ops.ACT_RELU, want_f32=False, want_bf=True)))
    mfma(f"mi_dense_bwd_dx_bf16[{M},{K},{N}]", fl,
         timed(lambda: ops.dense_bwd_dx_bf16(dz, w_bf, x, ops.ACT_RELU, K, N, want_f32=False, want_bf=True)))
    mfma(f"mi_dense_bwd_dw_bf16[{M},{K},{N}]", fl,
         timed(lambda: ops.dense_bwd_dw_bf16(x, dz, gw, None, accumulate=False)))

for M in [1024, 4096]:
    dims = [5, 256, 256, 1]
    xs = torch.randn(M, 5, device=dev)
    wts, bs = [], []
    for l in range(3):
        K, N = dims[l], dims[l + 1]
        w = torch.randn(K, N, device=dev) / K ** 0.5
        w_bf = torch.zeros(K, ops.pad8(N), dtype=torch.bfloat16, device=dev)
        wt_bf = torch.zeros(N, ops.pad8(K), dtype=torch.bfloat16, device=dev)
        ops.weights_to_bf16(w, w_bf, wt_bf)
        wts.append(wt_bf)
        bs.append(torch.zeros(N, device=dev))
    acts = [ops.ACT_RELU, ops.ACT_RELU, ops.ACT_NONE]
    mfma(f"mi_mlp_fwd_bf16(infer)[{M},5-256-256-1]", 2.0 * M * (5 * 256 + 256 * 256 + 256),
         timed(lambda: ops.mlp_fwd_bf16(xs, wts, bs, dims, acts, train=False)))

print(json.dumps(out, indent=1))
