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
def emulated_device(mlp_arithmetic="fp32"):
    """`mlp_arithmetic`: ops.mlp_arithmetic inside the context (the long host-level tests stay on the fused fp32 kernels,
    which the interpreter runs fastest; tests/test_emu_mlp_h3.py covers the resident kernels)."""
    from scnerf_amd import _capi, ops
    from tests.emu import harness
    saved = (_capi._lib, _capi.on_device, _capi.current_stream)
    saved_mode = ops.mlp_arithmetic()
    ops.mlp_arithmetic(mlp_arithmetic)
    ops._canon_cache.clear()
    _capi._lib = harness.lib()
    _capi.on_device = lambda t: True
    _capi.current_stream = lambda: None
    ops._index_cache.clear()
    ops._wgrad_ws.clear()
    ops._h3_tables.clear()
    try:
        yield
    finally:
        ops.mlp_arithmetic(saved_mode)
        ops._canon_cache.clear()
        _capi._lib, _capi.on_device, _capi.current_stream = saved
        ops._index_cache.clear()
        ops._wgrad_ws.clear()
        ops._h3_tables.clear()
