"""Summarise `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of bench.py into
per-kernel HBM traffic per launch (MI355X_MICROARCH.md, "HBM" and "rocprofv3 PMC slots":
the two counters do not fit one pass; FETCH_SIZE is in KiB and on gfx950 tallies 128-B
read requests at 64 B, so it is doubled; WRITE_SIZE is in KiB and exact for 16-B stores).

usage: pmc_summary.py <fetch_dir> <write_dir> <out.json>
"""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = acc[r["Kernel_Name"]]
            k[0] += float(r["Counter_Value"])
            k[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f_kib, n = fetch.get(k, (0.0, 0))
        w_kib, _ = write.get(k, (0.0, 0))
        out[short(k)] = {
            "launches": n,
            "fetch_bytes_per_launch": round(2.0 * f_kib * 1024),   # gfx950: x2
            "write_bytes_per_launch": round(w_kib * 1024),
            "hbm_bytes_per_launch": round(2.0 * f_kib * 1024 + w_kib * 1024),
        }
    import os
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    sys.path.insert(0, str(root))
    from nnx_ppo_amd.csrc.build import source_signature

    # the GPU box has no .git: the caller exports GIT_HEAD (tools/collect_profiles.sh)
    head = os.environ.get("GIT_HEAD")
    if not head:
        r = subprocess.run(["git", "-C", str(root), "rev-parse", "--short", "HEAD"],
                           capture_output=True, text=True)
        head = r.stdout.strip() if r.returncode == 0 else None
    json.dump({"git_head": head, "kernel_signature": source_signature(),
               "_note": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), KiB -> bytes; "
                        "separate --pmc passes; Infinity-Cache hits are counted",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(k[:70], v)


if __name__ == "__main__":
    main()
