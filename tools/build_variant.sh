#!/bin/bash
# Build ab/libmippo_<name>.so from the current objects with ONE source file taken from
# a git revision:  tools/build_variant.sh <name> <rev> <file.hip>
# Run with MIPPO_LIB=ab/libmippo_<name>.so to time it against the working tree on the
# same GPU box.
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; rev=$2; file=$3
mkdir -p ab /tmp/ab_$name
git show "$rev:nnx_ppo_amd/csrc/$file" > /tmp/ab_$name/$file
python -m nnx_ppo_amd.csrc.build > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Innx_ppo_amd/csrc \
  -c /tmp/ab_$name/$file -o /tmp/ab_$name/${file%.hip}.o
objs=$(ls nnx_ppo_amd/csrc/build/*.o | grep -v "/${file%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libmippo_$name.so $objs /tmp/ab_$name/${file%.hip}.o
echo ab/libmippo_$name.so
