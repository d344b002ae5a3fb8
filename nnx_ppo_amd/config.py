"""Process-wide compute settings.

`compute_dtype` selects the MFMA path of the Dense layers:
  "f32"  — v_mfma_f32_32x32x2_f32, exact fp32 products (tight-tolerance twin of
           the reference's fp32 XLA dots; what the parity tests pin);
  "bf16" — v_mfma_f32_16x16x32_bf16 with fp32 accumulation, fp32 master weights
           and fp32 sampler / loss / GAE / Adam / normaliser (BASELINE.json
           configs[1] is quoted in bf16).
"""
from __future__ import annotations

import contextlib

_state = {"compute_dtype": "f32"}


def compute_dtype() -> str:
    return _state["compute_dtype"]


def set_compute_dtype(name: str) -> None:
    if name not in ("f32", "bf16"):
        raise ValueError(f"compute dtype must be 'f32' or 'bf16', got {name!r}")
    _state["compute_dtype"] = name


@contextlib.contextmanager
def use_compute_dtype(name: str):
    prev = compute_dtype()
    set_compute_dtype(name)
    try:
        yield
    finally:
        set_compute_dtype(prev)
