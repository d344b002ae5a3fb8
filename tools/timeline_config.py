"""Timeline of one hip-graph iteration of a BASELINE config from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace -d DIR -o tr --output-format csv -- python3 tools/bench_configs.py c4
    python3 tools/timeline_config.py DIR [marker-substring]
Prints, for the LAST complete iteration (from one rollout kernel to the next), every kernel with
its start relative to the iteration, its duration, the gap to the previous kernel's end and
whether it overlapped another kernel — where a config's critical path goes."""
import csv
import sys
from pathlib import Path


def _short(name: str) -> str:
    """`void (anonymous namespace)::kernel<args>(params)` -> `kernel<args>` (clipped)."""
    import re

    m = re.search(r"([A-Za-z_][A-Za-z_0-9]*)(<[^(]*>)?\(", name.replace("(anonymous namespace)::", ""))
    if not m:
        return name[:70]
    return (m.group(1) + (m.group(2) or ""))[:70]


def main():
    d = Path(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else "rollout"
    f = next(d.rglob("*kernel_trace.csv"))
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 3:
        raise SystemExit(f"fewer than 3 kernels matching {marker!r}")
    a, b = marks[-3], marks[-2]
    it = rows[a:b]
    t0 = it[0][0]
    busy_end = t0
    total_gap = 0.0
    union = 0.0
    print(f"iteration: {(rows[b][0] - t0) / 1e3:.1f} us, {len(it)} kernels")
    for s, e, name in it:
        gap = (s - busy_end) / 1e3
        if gap > 0:
            total_gap += gap
        union += max(0, e - max(s, busy_end)) / 1e3
        short = _short(name)
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  gap {gap:7.1f}  {short}")
        busy_end = max(busy_end, e)
    dur = sum(e - s for s, e, _ in it) / 1e3
    print(f"sum of kernel durations {dur:.1f} us, union (GPU busy) {union:.1f} us, idle gaps "
          f"{total_gap:.1f} us")


if __name__ == "__main__":
    main()
