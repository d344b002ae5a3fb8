"""Phase stamps of the GRU sequence forward's time loop (tools/build_gru_trace.py):
    MIPPO_LIB=ab/libmippo_grutrace.so python tools/trace_gru.py
Cycles (s_memtime) per phase of one step of wave 0 of workgroup 7, median over steps 8..23, for
the training forms at T = 30, B = 1024, H = 64: plain (gi from memory), tail (head + sampler
behind the loop) and proj + tail (the input projection inside)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from nnx_ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
T, B, H, A2 = 30, 1024, 64, 2
g = torch.Generator().manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
gi, w_h, b, h0 = r(T, B, 3 * H), r(H, 3 * H) / H ** 0.5, r(H), r(B, H)
done = (torch.rand(T, B, generator=g) < 0.1).to(dev)
# stamps in program order: 0 start, 7 operand reads + sweep issued, 1 projection, 2 h MFMAs
# issued, 3 gate math, 4 LDS writes, 5 refill, 6 barrier
ORDER = [0, 7, 1, 2, 3, 4, 5, 6]
NAMES = ["reads + sweep", "proj MFMAs", "h MFMAs", "gate math", "LDS writes", "refill", "barrier"]


def report(tag, h_final):
    st = h_final[28:32].contiguous().view(torch.int64).flatten()[:128].view(16, 8).cpu().double()
    st = st[:, ORDER]
    d = st[:, 1:] - st[:, :-1]
    whole = (st[1:, 0] - st[:-1, 0]).median().item()
    gap = (st[1:, 0] - st[:-1, 7]).median().item()
    print(tag, "  ".join(f"{n} {d[:, i].median().item():.0f}" for i, n in enumerate(NAMES)),
          f"  step->step {gap:.0f}   whole step {whole:.0f} cycles")


for _ in range(3):
    out = ops.gru_seq_fwd(gi, w_h, b, h0, done, True, True)
torch.cuda.synchronize()
report("plain    :", out[3])
from nnx_ppo_amd.networks import dense_chain, factories  # noqa: E402
from nnx_ppo_amd.networks.types import Rngs  # noqa: E402
from nnx_ppo_amd import config  # noqa: E402

with config.use_compute_dtype("bf16"):
    net = factories.make_gru_actor_critic(5, 1, H, [256, 256], Rngs(3))
    net.to(dev)
    actor = net.layers[1].action
    rec, head, samp = actor.layers[1], actor.layers[2], actor.layers[3]
    dense_chain.refresh([head, rec._proj()])
    ex = r(T * B, A2 // 2)
    for _ in range(3):
        out = ops.gru_seq_fwd_tail(gi, rec.w_h.data, rec.b_hn.data, h0, done, head._ff,
                                   head.bias.data, A2, ex, samp._state(dev), 0, **samp._kw())
    torch.cuda.synchronize()
    report("tail     :", out[3])
    y_bf = torch.relu(r(T * B, H)).to(torch.bfloat16)
    for _ in range(3):
        out = ops.gru_seq_fwd_proj_tail(y_bf, rec._proj()._ff, rec.b_i.data, rec.w_h.data,
                                        rec.b_hn.data, h0, done, head._ff, head.bias.data, A2, ex,
                                        samp._state(dev), 0, T, **samp._kw())
    torch.cuda.synchronize()
    report("proj+tail:", out[3])
