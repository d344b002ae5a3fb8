"""Per-kernel averages of SQ counters from rocprofv3 `--pmc` passes of bench.py
(MI355X_MICROARCH.md, "rocprofv3 PMC slots": at most 8 SQ counters per pass; SQ_WAVE_CYCLES,
SQ_WAIT_* and SQ_ACTIVE_INST_* count quad-cycles).

usage: pmc_sq_summary.py <out.json> <pass_dir> [<pass_dir> ...]
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)


def main():
    out = collections.defaultdict(dict)
    for d in sys.argv[2:]:
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        assert files, f"no counter_collection.csv under {d}"
        acc = collections.defaultdict(lambda: [0.0, 0])
        for f in files:
            for r in csv.DictReader(open(f)):
                k = acc[(short(r["Kernel_Name"]), r["Counter_Name"])]
                k[0] += float(r["Counter_Value"])
                k[1] += 1
        for (kern, ctr), (tot, n) in acc.items():
            out[kern][ctr] = round(tot / n, 1)
            out[kern]["launches"] = n
    keep = {k: v for k, v in out.items()
            if k.startswith(("policy_", "tn_gemm_dw", "mlp_chain", "adam", "gae", "ppo_loss"))}
    for k, v in keep.items():
        w = v.get("SQ_WAVES")
        if w:
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS",
                      "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if c in v:
                    v[c + "_per_wave"] = round(v[c] / w, 1)
        wc = v.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if c in v:
                    v[c + "_frac_of_wave_cycles"] = round(v[c] / wc, 3)
    json.dump({"note": "averages per launch over all XCDs' SQs as rocprofv3 reports them; "
                       "per-wave = counter / SQ_WAVES", "kernels": keep},
              open(sys.argv[1], "w"), indent=1, sort_keys=True)
    for k, v in sorted(keep.items()):
        print(k, {c: v[c] for c in sorted(v) if c.endswith(("per_wave", "cycles"))})


main()
