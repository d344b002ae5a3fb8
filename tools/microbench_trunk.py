"""Whole-trunk kernels vs the per-layer path, forward and backward, in isolation."""
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timed(fn, reps=30):
    """us per call inside a captured HIP graph of `reps` back-to-back calls (the
    iteration runs as a graph: eager launches have a ~13 us floor that hides kernel
    changes of a few us)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3  # us


def shadows(w):
    K, N = w.shape
    w_bf = torch.zeros(K, ops.pad8(N), dtype=BF, device=dev)
    wt_bf = torch.zeros(N, ops.pad8(K), dtype=BF, device=dev)
    nf, nb = ops.frag_sizes(K, N)
    ff, fb = torch.zeros(nf, dtype=BF, device=dev), torch.zeros(nb, dtype=BF, device=dev)
    ops.weights_to_bf16_multi([w], [w_bf], [wt_bf], [ff], [fb])
    return w_bf, wt_bf, ff, fb


out = []
for name, dims in [("actor", [5, 64, 64, 64, 64, 2]), ("critic", [5, 256, 256, 1]),
                   ("c3actor", [17, 256, 256, 256, 256, 12]), ("c3critic", [17, 512, 512, 1])]:
    L = len(dims) - 1
    acts = [ops.ACT_RELU] * (L - 1) + [ops.ACT_NONE]
    ws = [torch.randn(dims[l], dims[l + 1], device=dev) / dims[l] ** 0.5 for l in range(L)]
    sh = [shadows(w) for w in ws]
    bs = [torch.zeros(dims[l + 1], device=dev) for l in range(L)]
    for M in [1024, 4096, 30720]:
        x = torch.randn(M, dims[0], device=dev)
        g = torch.randn(M, dims[-1], device=dev)
        ffs = [s[2] for s in sh]
        fbs = [s[3] for s in sh]
        rec = {"chain": name, "M": M, "rt": os.environ.get("MIPPO_TRUNK_RT", "auto")}
        rec["trunk_fwd_infer_us"] = round(timed(lambda: ops.mlp_fwd_bf16(x, ffs, bs, dims, acts, train=False)), 1)
        rec["trunk_fwd_train_us"] = round(timed(lambda: ops.mlp_fwd_bf16(x, ffs, bs, dims, acts, train=True)), 1)
        _, saved = ops.mlp_fwd_bf16(x, ffs, bs, dims, acts, train=True)
        auxs = [sv[1] for sv in saved]
        rec["trunk_bwd_dx_us"] = round(timed(lambda: ops.mlp_bwd_dx_bf16(g, None, ops.ACT_NONE, fbs, dims, acts, auxs, False)), 1)

        def per_layer_fwd():
            xb = ops.cast_pad_bf16(x)
            for l in range(L):
                _, xb, _ = ops.dense_fwd_bf16(xb, sh[l][1], bs[l], dims[l], dims[l + 1], acts[l],
                                              want_f32=l == L - 1, want_bf=True)

        def per_layer_bwd():
            dz = ops.cast_pad_bf16(g)
            for l in range(L - 1, 0, -1):
                _, dz = ops.dense_bwd_dx_bf16(dz, sh[l][0], auxs[l - 1], acts[l - 1], dims[l],
                                              dims[l + 1], want_f32=False, want_bf=True)

        rec["per_layer_fwd_us"] = round(timed(per_layer_fwd), 1)
        rec["per_layer_bwd_dx_us"] = round(timed(per_layer_bwd), 1)
        out.append(rec)
        print(rec, flush=True)
