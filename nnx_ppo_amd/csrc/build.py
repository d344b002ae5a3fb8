"""Build libmippo.so (gfx950) in-tree with hipcc.

    python -m nnx_ppo_amd.csrc.build [--force]

Every `*.hip` in this directory is compiled to an object (in parallel, skipped
when up to date) and linked into `nnx_ppo_amd/libmippo.so`.  No torch headers,
no cmake: the library only depends on the HIP runtime, and hipcc cross-compiles
gfx950 code objects without a GPU present.
"""
from __future__ import annotations

import concurrent.futures
import os
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent
PKG = CSRC.parent
ROOT = PKG.parent
LIB = PKG / "libmippo.so"
OBJ = CSRC / "build"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-Wall",
    "-Wno-unused-function",
    f"-I{ROOT / 'include'}",
    f"-I{CSRC}",
]


def _newest_header() -> float:
    hs = list(CSRC.glob("*.h")) + list((ROOT / "include").glob("*.h"))
    return max(h.stat().st_mtime for h in hs)


def _compile(src: Path, force: bool, hdr_mtime: float) -> Path:
    obj = OBJ / (src.stem + ".o")
    if (
        not force
        and obj.exists()
        and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_mtime)
    ):
        return obj
    cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def source_signature() -> str:
    """sha256 over every kernel source and header (sorted by name): the identity of the
    kernels a profile was taken on.  `tools/pmc_summary.py` stamps it into the PMC traffic
    file; `bench.py` marks `roofline.traffic` stale when it no longer matches."""
    import hashlib

    h = hashlib.sha256()
    files = sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) +
                   list((ROOT / "include").glob("*.h")), key=lambda f: f.name)
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def build(force: bool = False, jobs: int | None = None) -> Path:
    OBJ.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError("no .hip sources found")
    hdr = _newest_header()
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) - 1))
    with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr), srcs))
    if (
        not force
        and LIB.exists()
        and all(LIB.stat().st_mtime > o.stat().st_mtime for o in objs)
    ):
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)]
    cmd += [str(o) for o in objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
