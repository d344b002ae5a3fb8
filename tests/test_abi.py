"""The C-ABI library loads (no GPU needed) and exports exactly the symbols
include/mippo.h declares; the ctypes binding covers each of them."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    text = (ROOT / "include" / "mippo.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    from nnx_ppo_amd.csrc.build import build

    return build()


def test_header_symbols_exported(built):
    lib = ctypes.CDLL(str(built))
    names = _declared()
    assert "mi_gae_f32" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mippo.h but not exported"


def test_binding_covers_header(built):
    from nnx_ppo_amd import _lib

    assert sorted(_lib.exported_symbols()) == _declared()
    assert _lib.lib().mi_abi_version() == _lib.ABI_VERSION
    assert _lib.last_error() == ""


def test_cpu_tensor_is_refused(built):
    import torch

    from nnx_ppo_amd import _lib

    with pytest.raises(_lib.MippoError):
        _lib.ptr(torch.zeros(4))


def test_product_does_not_import_oracle():
    bad = []
    for p in (ROOT / "nnx_ppo_amd").rglob("*.py"):
        s = p.read_text()
        if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M):
            bad.append(str(p))
    assert not bad, f"product files import the oracle: {bad}"
