"""TF/s of the per-layer NT GEMM (mi_dense_fwd_bf16 / mi_dense_bwd_dx_bf16) at BASELINE config
3's layer shapes, with the 256 x 256 direct-to-LDS kernel and (MIPPO_GEMM256=0) without it,
and of torch.mm (hipBLASLt) on the same operands — replays of a HIP graph, random data.

    python tools/microbench_gemm256.py            # the new kernel
    MIPPO_GEMM256=0 python tools/microbench_gemm256.py
"""
import json
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops  # noqa: E402


def timed(fn, iters=30):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3  # us


def main():
    dev = torch.device("cuda", 0)
    out = []
    for M, K, N in [(61440, 512, 512), (61440, 256, 256), (61440, 32, 512), (30720, 256, 256),
                    (30720, 512, 512)]:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(K, N, device=dev) / math.sqrt(K)
        b = torch.randn(N, device=dev)
        x_bf = ops.cast_pad_bf16(x)
        w_bf = torch.zeros(K, N, dtype=torch.bfloat16, device=dev)
        wt_bf = torch.zeros(N, K, dtype=torch.bfloat16, device=dev)
        ops.weights_to_bf16(w, w_bf, wt_bf)
        dz_bf = ops.cast_pad_bf16(torch.randn(M, N, device=dev))
        prev_bf = ops.cast_pad_bf16(torch.randn(M, K, device=dev))
        fl = 2.0 * M * K * N
        t_f = timed(lambda: ops.dense_fwd_bf16(x_bf, wt_bf, b, K, N, ops.ACT_RELU,
                                               want_f32=False, want_bf=True))
        t_d = timed(lambda: ops.dense_bwd_dx_bf16(dz_bf, w_bf, prev_bf, ops.ACT_RELU, K, N,
                                                  want_f32=False, want_bf=True))
        xb, wb = x_bf[:, :K].contiguous(), w_bf[:, :N].contiguous()
        t_b = timed(lambda: torch.mm(xb, wb))
        rec = {"M": M, "K": K, "N": N, "fwd_us": round(t_f, 1), "fwd_TFs": round(fl / t_f / 1e6, 1),
               "dx_us": round(t_d, 1), "dx_TFs": round(fl / t_d / 1e6, 1),
               "torch_mm_us": round(t_b, 1), "torch_mm_TFs": round(fl / t_b / 1e6, 1)}
        print(json.dumps(rec), flush=True)
        out.append(rec)


if __name__ == "__main__":
    main()
