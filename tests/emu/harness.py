"""TEST INFRASTRUCTURE: loads tests/emu/_build/libscnerf_emu.so (the product's kernel
sources compiled for the CPU SIMT interpreter) and calls its C ABI on numpy buffers.
Used by the `-m "not gpu"` logic tests only."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from scnerf_amd import _capi  # noqa: E402  (prototype table only; no library is loaded by it)

_lib = None


def lib():
    global _lib
    if _lib is None:
        sys.path.insert(0, HERE)
        import build_emu
        path = build_emu.build()
        _lib = _capi.bind(ctypes.CDLL(path))
    return _lib


def ptr(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"], "need a contiguous ndarray"
    return a.ctypes.data_as(ctypes.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def call(name, *args):
    st = getattr(lib(), name)(*[ptr(a) if isinstance(a, np.ndarray) or a is None else a for a in args])
    assert st == 0, "%s returned %d" % (name, st)


def lib_call_status(name, *args):
    """as call(), returning the status instead of asserting it"""
    return getattr(lib(), name)(*[ptr(a) if isinstance(a, np.ndarray) or a is None else a for a in args])
