"""GRU sequence kernels in isolation (in-graph timing): fp32 VALU vs bf16 matrix cores."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=10):
    for _ in range(2):
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
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


for T, B, H in [(30, 1024, 64), (1, 4096, 64), (30, 1024, 128)]:
    gi = torch.randn(T, B, 3 * H, device=dev)
    w_h = torch.randn(H, 3 * H, device=dev) / H ** 0.5
    b = torch.zeros(H, device=dev)
    h0 = torch.randn(B, H, device=dev)
    done = torch.rand(T, B, device=dev) < 0.1
    g_h = torch.randn(T, B, H, device=dev)
    rec = {"T": T, "B": B, "H": H}
    for mfma in (False, True):
        tag = "mfma" if mfma else "f32"
        rec[f"fwd_infer_{tag}_us"] = round(timed(lambda: ops.gru_seq_fwd(gi, w_h, b, h0, done, False, mfma)), 1)
        rec[f"fwd_train_{tag}_us"] = round(timed(lambda: ops.gru_seq_fwd(gi, w_h, b, h0, done, True, mfma)), 1)
        _, hp, gates, _ = ops.gru_seq_fwd(gi, w_h, b, h0, done, True, mfma)
        rec[f"bwd_{tag}_us"] = round(timed(lambda: ops.gru_seq_bwd(g_h, gates, hp, w_h, done, mfma)), 1)
    print(rec, flush=True)
