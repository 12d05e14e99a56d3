"""TEST INFRASTRUCTURE: runs the package's *host layer* (autograd functions, render / create_nerf /
optimizer mirrors) on CPU tensors against the CPU SIMT interpreter build of the kernels
(tests/emu/_build/libscnerf_emu.so: the product's .hip sources, same C ABI, host pointers).

Only tests use this; the product never looks for the emulator.  Inside the context
    _capi.load()            -> the emulator library
    _capi.on_device(t)      -> True for every tensor
    _capi.current_stream()  -> NULL
and everything is restored on exit."""
import contextlib


@contextlib.contextmanager
def emulated_device():
    from scnerf_amd import _capi, ops
    from tests.emu import harness
    saved = (_capi._lib, _capi.on_device, _capi.current_stream)
    _capi._lib = harness.lib()
    _capi.on_device = lambda t: True
    _capi.current_stream = lambda: None
    ops._index_cache.clear()
    ops._wgrad_ws.clear()
    try:
        yield
    finally:
        _capi._lib, _capi.on_device, _capi.current_stream = saved
        ops._index_cache.clear()
        ops._wgrad_ws.clear()
