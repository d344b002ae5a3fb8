#!/bin/bash
# SQ counter passes of bench.py on a GPU box (run through gpurun from the repo root):
#   tools/collect_pmc_sq.sh <tag>   ->  gpurun_out/<tag>_pmc_sq.json
# Counters go in their own rocprofv3 runs (--pmc only, at most 8 SQ counters per pass:
# MI355X_MICROARCH.md, rocprofv3 PMC slots); the program itself follows `--`.
set -eo pipefail
tag=${1:-build}
root=$(pwd)
out=$root/gpurun_out/${tag}_sq
mkdir -p "$out"
export TMPDIR=/tmp
cmd="python3 bench.py --no-cpu-baseline --no-train-ppo --steps 3 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SMEM \
    SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$out/p1" -o p1 --output-format csv -- $cmd > /dev/null
echo "pass 1 done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \
    SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d "$out/p2" -o p2 \
    --output-format csv -- $cmd > /dev/null
echo "pass 2 done"
dirs="$out/p1 $out/p2"
# the MFMA-busy counter on its own: a box whose rocprofv3 does not know it still gives the rest
if rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d "$out/p3" -o p3 \
    --output-format csv -- $cmd > /dev/null 2> "$out/p3.err"; then
  dirs="$dirs $out/p3"
  echo "pass 3 done"
else
  echo "pass 3 (SQ_VALU_MFMA_BUSY_CYCLES) not available: $(tail -1 "$out/p3.err")"
fi
python3 tools/pmc_sq_summary.py "$root/gpurun_out/${tag}_pmc_sq.json" $dirs
rm -rf "$out"
echo "$root/gpurun_out/${tag}_pmc_sq.json"
