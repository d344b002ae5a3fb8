"""ctypes binding of libmippo.so (the C ABI declared in include/mippo.h).

The product path has no fallback: if the shared library is missing or a symbol
cannot be resolved, importing an op raises.  Tensors cross the boundary as raw
device pointers (`tensor.data_ptr()`), sizes as int64, and the HIP stream as
the handle of torch's current stream.
"""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_uint64, c_void_p
from pathlib import Path

import torch

# MIPPO_LIB points the binding at another build of the same ABI (A/B timing of kernel
# variants on one GPU box: boxes differ by more than most kernel changes do)
_LIB_PATH = Path(os.environ.get("MIPPO_LIB") or Path(__file__).resolve().parent / "libmippo.so")

ABI_VERSION = 1

_HEADER = Path(__file__).resolve().parent.parent / "include" / "mippo.h"

_CTYPES = {
    "float": c_float,
    "double": c_double,
    "int": c_int,
    "int64_t": c_int64,
    "uint64_t": c_uint64,
    "mi_stream_t": c_void_p,
}


def _parse_header(path: Path) -> dict[str, tuple]:
    """Read the C ABI from include/mippo.h so the binding can never drift from
    the header: name -> (restype, argtypes)."""
    text = re.sub(r"/\*.*?\*/", "", path.read_text(), flags=re.S)
    sigs: dict[str, tuple] = {}
    for m in re.finditer(r"([A-Za-z_0-9]+[\s\*]+)(mi_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        restype = c_char_p if "char" in ret else _CTYPES[ret.replace("const", "").strip()]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(c_void_p)
                else:
                    ty = a.replace("const", "").split()[0]
                    argtypes.append(_CTYPES[ty])
        sigs[name] = (restype, argtypes)
    return sigs


_SIGNATURES = _parse_header(_HEADER)


class MippoError(RuntimeError):
    pass


def _load() -> ctypes.CDLL:
    if not _LIB_PATH.exists():
        raise MippoError(
            f"{_LIB_PATH} not found: build it with `python -m nnx_ppo_amd.csrc.build` "
            "(there is no CPU / PyTorch fallback for the HIP path)"
        )
    lib = ctypes.CDLL(os.fspath(_LIB_PATH))
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = restype
    got = lib.mi_abi_version()
    if got != ABI_VERSION:
        raise MippoError(f"libmippo ABI {got} != binding ABI {ABI_VERSION}; rebuild")
    return lib


class Profiler:
    """Per-entry-point device timing: when active, every C-ABI call is bracketed by
    HIP events recorded on torch's current stream (the stream the kernels are
    enqueued on).  `summary()` synchronises and returns, per symbol, the call
    count, total / average milliseconds and the integer arguments of each call
    (shapes), from which bench.py derives algorithmic FLOPs / bytes."""

    def __init__(self):
        self.active = False
        self.records: list = []
        self.next_flops = None  # set by a wrapper that knows the work of its next call
        self.next_bytes = None  # likewise: algorithmic HBM bytes of the next call

    def __enter__(self):
        self.records = []
        self.active = True
        return self

    def __exit__(self, *exc):
        self.active = False
        return False

    def summary(self) -> dict:
        torch.cuda.synchronize()
        out: dict = {}
        for name, ints, e0, e1, flops, nbytes in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "args": [], "flops": 0.0,
                                      "work": []})
            d["calls"] += 1
            ms = e0.elapsed_time(e1)
            d["ms"] += ms
            d["args"].append((ints, ms))
            d["work"].append((flops, nbytes))
            if flops is not None:
                d["flops"] += flops
        for d in out.values():
            d["avg_ms"] = d["ms"] / max(d["calls"], 1)
        return out


profiler = Profiler()


class _Proxy:
    """Attribute access returns the raw ctypes function, or an event-bracketed
    wrapper of it while the profiler is active."""

    def __init__(self, cdll: ctypes.CDLL):
        self._cdll = cdll

    def __getattr__(self, name: str):
        fn = getattr(self._cdll, name)
        if not profiler.active:
            return fn

        def timed(*args):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            ints = tuple(a for a in args if isinstance(a, int) and not isinstance(a, bool) and 0 <= a < (1 << 40))
            profiler.records.append((name, ints, e0, e1, profiler.next_flops,
                                     profiler.next_bytes))
            profiler.next_flops = None
            profiler.next_bytes = None
            return rc

        return timed


_lib: _Proxy | None = None


def lib() -> _Proxy:
    global _lib
    if _lib is None:
        _lib = _Proxy(_load())
    return _lib


def exported_symbols() -> list[str]:
    return sorted(_SIGNATURES)


def last_error() -> str:
    return lib().mi_last_error().decode()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise MippoError(f"{what} failed (rc={rc}): {last_error()}")


def stream() -> int:
    """Handle of torch's current HIP stream (what kernels are enqueued on)."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t: torch.Tensor | None, dtype: torch.dtype | None = None) -> int | None:
    """Device pointer of a contiguous GPU tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MippoError(
            "libmippo ops need GPU tensors (got a CPU tensor); there is no CPU fallback"
        )
    if not t.is_contiguous():
        raise MippoError("libmippo ops need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise MippoError(f"expected dtype {dtype}, got {t.dtype}")
    return t.data_ptr()
