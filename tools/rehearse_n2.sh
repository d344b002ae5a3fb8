#!/bin/bash
# Two ranks of bench.py on a ONE-GPU box (every rank on cuda:0, rendezvous through gloo, the
# exchanges through the one-shot peer kernels over IPC): a rehearsal of the N > 1 code path, not
# a scaling measurement — the two ranks share one GPU.
#   tools/rehearse_n2.sh [port]
set -eo pipefail
port=${1:-29511}
MIPPO_DIST_BACKEND=gloo MIPPO_SINGLE_DEVICE=1 timeout -k 10 400 python3 -m torch.distributed.run \
  --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port "$port" \
  bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-train-ppo \
  > gpurun_out/bench_n2.json 2> gpurun_out/bench_n2.err || { tail -20 gpurun_out/bench_n2.err; exit 1; }
python3 - <<'P'
import json
d = json.loads(open("gpurun_out/bench_n2.json").read().strip().splitlines()[-1])
print(d["value"], "env-steps/s", d["ms_per_step"], "ms/iter", d["config"]["transport"],
      d.get("launch_mode"))
ks = d["kernels_ms_per_iter"]
for k, v in sorted(((k, v) for k, v in ks.items() if isinstance(v, dict)),
                   key=lambda kv: -kv[1]["ms"])[:12]:
    print(f"  {k:44s} {v['calls']:4d} x {1e3 * v['ms'] / v['calls']:7.2f} us")
P
