"""The training-size policy kernels (replay forward with the bootstrap tail, backward) of
the C2 network under one instantiation (MIPPO_POLICY_SHAPE, read once per process), or —
with `--sweep` — every instantiation of the menu, one child process each.

    MIPPO_LIB=ab/libmippo_pv.so python tools/microbench_policy.py --sweep
"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

MENU = ["", "16,1,4,4,0,2", "4,4,4,4,0,2", "16,1,4,4,1,2", "16,1,4,4,1,3", "16,1,4,4,0,3",
        "8,1,2,4,1,4", "8,1,3,4,1,3", "16,1,6,4,1,2", "12,1,6,4,1,2", "8,1,4,4,1,3"]


def one():
    import torch

    from nnx_ppo_amd import config
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import PPONetworkOutput, Rngs
    from nnx_ppo_amd.optim import Optimizer
    from tools._timing import timed

    dev = torch.device("cuda:0")
    config.set_compute_dtype("bf16")
    T, B = 30, 1024
    net = factories.make_mlp_actor_critic(5, 1, [64] * 4, [256] * 2, Rngs(17))
    net.to(dev)
    opt = Optimizer(net, 1e-4, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    obs, last = rn(T, B, 5), rn(B, 5)
    done = torch.zeros(T, B, dtype=torch.bool, device=dev)
    extras = [obs, {"action": [None] * 5 + [rn(T, B, 1)], "value": [None] * 3}]
    st = net.initialize_state(B)
    box = {}

    def fwd():
        box["r"] = net.replay_with_bootstrap(st, obs, done, extras, last)

    fwd()
    ctx = box["r"][0]
    g_out = PPONetworkOutput(None, rn(T, B), rn(T, B))

    def bwd():
        opt.begin()
        net.replay_backward(ctx, g_out, 1.0 / (T * B))

    # the backward launches policy_bwd + the grouped dW; time the policy part alone too
    from nnx_ppo_amd import _lib

    def split():
        torch.cuda.synchronize()
        with _lib.profiler as prof:
            fwd()
            bwd()
        s = prof.summary()
        return {k: round(v["ms"] * 1e3, 1) for k, v in s.items() if "workspace" not in k}

    rec = {"shape": os.environ.get("MIPPO_POLICY_SHAPE", "default"),
           "fwd_us": round(timed(fwd), 2), "bwd_plus_dw_us": round(timed(bwd), 2),
           "eager_events_us": split()}
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    if "--sweep" in sys.argv:
        for shape in MENU:
            env = dict(os.environ)
            if shape:
                env["MIPPO_POLICY_SHAPE"] = shape
            else:
                env.pop("MIPPO_POLICY_SHAPE", None)
            r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True,
                               timeout=300)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            err = r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else ""
            print(lines[-1] if lines else f"{shape or 'default'}: FAILED {err}", flush=True)
    else:
        one()
