#!/bin/bash
# ab/libmippo_trace.so: the working tree with mlp_bf16.hip / gemm_bf16.hip compiled -DMIPPO_TRACE
# (per-workgroup phase stamps, read by tools/trace_policy.py).
set -euo pipefail
cd "$(dirname "$0")/.."
mkdir -p ab /tmp/ab_trace
python -m nnx_ppo_amd.csrc.build > /dev/null
for f in mlp_bf16 gemm_bf16 trunk_ws; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMIPPO_TRACE -Iinclude -Innx_ppo_amd/csrc \
    -c nnx_ppo_amd/csrc/$f.hip -o /tmp/ab_trace/$f.o &
done
wait
objs=$(ls nnx_ppo_amd/csrc/build/*.o | grep -v "/mlp_bf16.o\|/gemm_bf16.o\|/trunk_ws.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libmippo_trace.so $objs /tmp/ab_trace/mlp_bf16.o /tmp/ab_trace/gemm_bf16.o /tmp/ab_trace/trunk_ws.o
echo ab/libmippo_trace.so
