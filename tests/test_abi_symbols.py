"""The C-ABI library loads (no GPU needed) and exports every function include/scnerf_hip.h
declares; the ctypes prototype table covers exactly that set."""
import ctypes
import os
import re

import pytest

from scnerf_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, "include", "scnerf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long long)\s+(scnerf_\w+)\s*\(", src)))


def test_header_and_prototype_table_agree():
    names = declared()
    assert len(names) >= 25
    table = sorted(list(_capi.PROTOTYPES) + list(_capi.SIZE_FUNCS))
    assert names == table


def test_product_library_exports_every_symbol():
    if not os.path.isfile(_capi.LIB_PATH):
        from scnerf_amd.csrc import build
        build.build(verbose=False)
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared():
        assert hasattr(lib, name), name
    assert lib.scnerf_abi_version() == 4


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_capi.ScnerfLibraryError):
        _capi.load()


def test_cpu_tensors_are_rejected():
    import torch
    from scnerf_amd import ops
    with pytest.raises(RuntimeError):
        ops.coarse_sample(torch.zeros(4, 11), torch.linspace(0, 1, 8), None, False)
