"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
that include/dino_hip.h declares (no compute calls: this container has no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "dino_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(dino_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = _declared()
    assert "dino_gemm" in names and "dino_loss_fwd_bwd" in names and len(names) >= 20


def test_library_exports_every_declared_symbol():
    from dino_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in dino_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == _declared()
    assert lib.dino_abi_version() == _lib.ABI_VERSION


def test_product_fails_loudly_on_cpu_tensors():
    import torch
    from dino_amd import ops
    from dino_amd._lib import DinoHipError
    with pytest.raises(DinoHipError):
        ops.gemm(torch.zeros(8, 8), torch.zeros(8, 8), torch.zeros(8, 8))
