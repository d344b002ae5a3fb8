#!/bin/bash
# ab/libmippo_<name>.so = the whole library as of git revision <rev> (A/B partner of the
# working tree on one GPU box: MIPPO_LIB=ab/libmippo_<name>.so python bench.py ...).
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; rev=$2
d=/tmp/ab_rev_$name
rm -rf $d; mkdir -p $d ab
git archive "$rev" nnx_ppo_amd/csrc include | tar -x -C $d
objs=""
for f in $d/nnx_ppo_amd/csrc/*.hip; do
  o=${f%.hip}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$d/include -I$d/nnx_ppo_amd/csrc -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libmippo_$name.so $objs
echo ab/libmippo_$name.so
