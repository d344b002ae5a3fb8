"""us of the THIN layers of BASELINE config 3's trunks through the per-layer entry points (first
layers K = 17, heads N = 1 / 12, and the heads' dX) at M = 61 440, with the bytes each must
move — the part of a layer-by-layer trunk that is bandwidth, not matrix-core, bound."""
import json
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from nnx_ppo_amd import ops  # noqa: E402
from microbench_gemm256 import timed  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    M = 61440
    for K, N, last in [(17, 512, False), (17, 256, False), (512, 1, True), (256, 12, True)]:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(K, N, device=dev) / math.sqrt(K)
        b = torch.randn(N, device=dev)
        x_bf = ops.cast_pad_bf16(x)
        w_bf = torch.zeros(K, ops.pad8(N), dtype=torch.bfloat16, device=dev)
        wt_bf = torch.zeros(N, ops.pad8(K), dtype=torch.bfloat16, device=dev)
        ops.weights_to_bf16(w, w_bf, wt_bf)
        act = ops.ACT_NONE if last else ops.ACT_RELU
        t_f = timed(lambda: ops.dense_fwd_bf16(x_bf, wt_bf, b, K, N, act, want_f32=last,
                                               want_bf=not last))
        rec = {"layer": f"{K}->{N}", "fwd_us": round(t_f, 1),
               "fwd_bytes_MB": round(M * (ops.pad8(K) * 2 + (N * 4 if last else ops.pad8(N) * 2))
                                     / 1e6, 1)}
        if last:  # the head's dX: gX [M, K] = dZ [M, N] W^T x relu'(prev [M, K])
            dz_bf = ops.cast_pad_bf16(torch.randn(M, N, device=dev))
            prev_bf = ops.cast_pad_bf16(torch.randn(M, K, device=dev))
            t_d = timed(lambda: ops.dense_bwd_dx_bf16(dz_bf, w_bf, prev_bf, ops.ACT_RELU, K, N,
                                                      want_f32=False, want_bf=True))
            rec["dx_us"] = round(t_d, 1)
            rec["dx_bytes_MB"] = round(M * (ops.pad8(N) * 2 + 2 * ops.pad8(K) * 2) / 1e6, 1)
        t_c = timed(lambda: ops.cast_pad_bf16(x))
        rec["cast_pad_us"] = round(t_c, 1)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
