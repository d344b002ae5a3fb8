#!/bin/bash
# Collect the judged artifacts of one build on a GPU box (run through gpurun from the repo root):
#   GIT_HEAD=$(git rev-parse --short HEAD) gpurun ... "GIT_HEAD=$GIT_HEAD tools/collect_profiles.sh <tag>"
# -> gpurun_out/<tag>/{bench.json, stats/, fetch/, write/, kernel_stats.csv, pmc_traffic.json}
# Kernel timing and PMC counters are separate rocprofv3 runs, and FETCH_SIZE / WRITE_SIZE
# separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -eo pipefail
tag=${1:-build}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench.json"
echo "bench done"
rocprofv3 --kernel-trace --stats -d "$out/stats" -o stats --output-format csv -- \
    python3 bench.py --no-cpu-baseline > "$out/bench_under_rocprof.json"
echo "stats done"
rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch --output-format csv -- \
    python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write --output-format csv -- \
    python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null
echo "write done"
python3 tools/pmc_summary.py "$out/fetch" "$out/write" "$out/pmc_traffic.json"
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
# the raw traces are large; keep the summaries only
rm -rf "$out/stats" "$out/fetch" "$out/write"
cat "$out/bench.json"
