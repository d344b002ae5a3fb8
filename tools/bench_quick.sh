#!/bin/bash
# Quick A/B on a GPU box: the default bench line without the CPU baseline and the train_ppo leg,
# reduced to (env-steps/s, ms/iteration) and the per-call times of the C-ABI entry points.
#   tools/bench_quick.sh <tag> [ENV=value ...]
set -eo pipefail
tag=${1:-q}; shift || true
for kv in "$@"; do export "$kv"; done
python3 bench.py --no-cpu-baseline --no-train-ppo > gpurun_out/$tag.json 2> gpurun_out/$tag.err
python3 - "$tag" <<'P'
import json, sys
d = json.loads(open(f"gpurun_out/{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"] / 1e6, 2), "M env-steps/s", d["ms_per_step"], "ms/iter")
ks = d.get("kernels_ms_per_iter") or {}
for k, v in sorted(((k, v) for k, v in ks.items() if isinstance(v, dict)), key=lambda kv: -kv[1]["ms"])[:9]:
    print(f"  {k:42s} {v['calls']:4d} x {1e3 * v['ms'] / v['calls']:7.2f} us = {v['ms']:.4f} ms")
P
