#!/bin/bash
# ab/libmippo_pv.so = the working tree's library with the whole menu of training-size
# policy-kernel instantiations (mlp_bf16.hip built with -DMIPPO_POLICY_VARIANTS);
# MIPPO_LIB=ab/libmippo_pv.so MIPPO_POLICY_SHAPE=... selects one (tools/microbench_policy.py).
set -euo pipefail
cd "$(dirname "$0")/.."
python -m nnx_ppo_amd.csrc.build > /dev/null
mkdir -p ab /tmp/ab_pv
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Innx_ppo_amd/csrc \
  -DMIPPO_POLICY_VARIANTS -c nnx_ppo_amd/csrc/mlp_bf16.hip -o /tmp/ab_pv/mlp_bf16.o
objs=$(ls nnx_ppo_amd/csrc/build/*.o | grep -v "/mlp_bf16.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libmippo_pv.so $objs /tmp/ab_pv/mlp_bf16.o
echo ab/libmippo_pv.so
