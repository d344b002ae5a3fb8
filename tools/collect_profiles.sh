#!/bin/bash
# Collect the judged artifacts of one build on a GPU box (run through gpurun from the repo root):
#   GIT_HEAD=$(git rev-parse --short HEAD) gpurun ... "GIT_HEAD=$GIT_HEAD tools/collect_profiles.sh <tag>"
# -> gpurun_out/<tag>/{bench.json, kernel_stats.csv, pmc_traffic.json, bench_under_rocprof.json}
# Kernel timing and PMC counters are separate rocprofv3 runs, and FETCH_SIZE / WRITE_SIZE
# separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots).  The profiled runs time the
# headline workload ONLY (--no-other-configs): the per-kernel averages of the stats file and of
# the PMC summary are then C2's, the ones bench.py's roofline object quotes (the dW kernel of
# C3 / C4 has the same name and other sizes).  The PMC summary is written FIRST and copied to
# profiles/<round>_pmc_traffic.json on the box, so that the bench line produced last reads the
# traffic of ITS OWN tree (traffic_stale false).
set -eo pipefail
tag=${1:-build}
round=${ROUND:-r03}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/stats" -o stats --output-format csv -- \
    python3 bench.py --no-cpu-baseline --no-other-configs > "$out/bench_under_rocprof.json"
echo "stats done"
rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch --output-format csv -- \
    python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 > /dev/null
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write --output-format csv -- \
    python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 > /dev/null
echo "write done"
python3 tools/pmc_summary.py "$out/fetch" "$out/write" "$out/pmc_traffic.json"
cp "$out/pmc_traffic.json" "$root/profiles/${round}_pmc_traffic.json"
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
# the raw traces are large; keep the summaries only
rm -rf "$out/stats" "$out/fetch" "$out/write"
python3 bench.py > "$out/bench.json"
echo "bench done"
cat "$out/bench.json"
