"""ctypes prototypes of the C ABI declared in include/scnerf_hip.h.

`load()` opens the in-tree libscnerf_hip.so (built by scnerf_amd.csrc.build) and fails
loudly when it is missing -- there is no CPU fallback in this package."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int, c_longlong, c_void_p, c_float, c_double

D = c_double

F = c_float

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCNERF_HIP_LIB") or os.path.join(_HERE, "libscnerf_hip.so")   # override: kernel experiments

P = c_void_p
I = c_int
LL = c_longlong

# name -> argument ctypes (every function returns int)
PROTOTYPES = {
    "scnerf_abi_version": [],
    "scnerf_searchsorted": [P, P, P, I, I, I, I, I, I, P],
    "scnerf_sample_pdf": [P, P, P, I, P, P, P, I, I, I, P],
    "scnerf_render_randoms": [ctypes.c_ulonglong, ctypes.c_ulonglong, P, LL, P, LL, P, LL, P, LL, F, P],
    "scnerf_coarse_sample": [P, I, P, P, P, P, I, I, I, P],
    "scnerf_fine_sample": [P, I, P, P, P, I, P, P, P, P, P, P, I, I, I, P],
    "scnerf_camera_rays_fwd": [P, P, I, P, I, P, P, F, I, P, P, F, I, P, F, P, F, I, I, I, I, P, P, I, P],
    "scnerf_camera_rays_bwd": [P, P, I, P, I, P, P, F, I, P, P, F, I, P, F, P, F, I, I, I, I, P, P, P, P, P, P, P, P, I, P],
    "scnerf_camera_matrices_fwd": [P, P, F, I, P, P, F, I, P, P, P],
    "scnerf_camera_matrices_bwd": [P, F, I, P, P, F, I, P, P, P, P, P],
    "scnerf_pinhole_rays": [P, I, P, F, I, I, P, P, I, P],
    "scnerf_ndc_fwd": [I, I, P, F, P, P, P, P, I, P],
    "scnerf_ndc_bwd": [I, I, P, F, P, P, P, P, P, P, P, I, P],
    "scnerf_pack_rays_fwd": [I, I, P, F, P, P, F, F, I, P, I, P],
    "scnerf_pack_rays_bwd": [I, I, P, F, P, P, I, P, P, P, P, I, P],
    "scnerf_upsample_grid_fwd": [P, F, I, I, I, I, P, P],
    "scnerf_upsample_grid_bwd": [P, F, I, I, I, I, P, P],
    "scnerf_prd_loss_fwd": [P, P, P, P, P, P, P, P, F, F, I, I, I, P, P, P, P],
    "scnerf_prd_loss_bwd": [P, P, P, P, P, P, P, P, F, F, I, I, P, P, P, P, P, P, P, P, P, P],
    "scnerf_embed_fwd": [P, LL, I, P, I, I, P, P],
    "scnerf_embed_bwd": [P, P, LL, I, P, I, I, P, P],
    "scnerf_prd_filter": [P, P, P, P, P, P, P, P, F, F, I, I, P, P],
    "scnerf_npp_intersect_fwd": [P, P, P, P, I, P],
    "scnerf_npp_intersect_bwd": [P, P, P, P, P, I, P],
    "scnerf_npp_perturb_fwd": [P, P, P, I, I, P],
    "scnerf_npp_perturb_bwd": [P, P, P, I, I, P],
    "scnerf_npp_sample_pdf": [P, P, P, P, P, P, P, I, I, I, P],
    "scnerf_npp_sample_pdf_bwd": [P, P, P, P, I, I, I, P],
    "scnerf_npp_points_fwd": [P, P, P, P, P, P, P, P, I, I, I, P],
    "scnerf_npp_points_bwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, P],
    "scnerf_npp_composite_fwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, P],
    "scnerf_npp_composite_bwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, P],
    "scnerf_npp_camera_rays_fwd": [P, P, I, P, P, P, F, I, P, P, F, I, P, F, P, F, I, I, I, I, P, P, I, P],
    "scnerf_npp_camera_rays_bwd": [P, P, I, P, P, P, F, I, P, P, F, I, P, F, P, F, I, I, I, I, P, P, P, P, P, P, P,
                                   P, P, I, P],
    "scnerf_adam_step": [P, P, P, P, LL, D, D, D, D, D, LL, P],
    "scnerf_adam_step_range": [P, P, P, P, LL, D, D, D, D, D, LL, LL, LL, P],
    "scnerf_composite_fwd": [P, P, P, I, P, I, P, P, P, P, P, I, I, P],
    "scnerf_composite_bwd": [P, P, P, I, P, I, P, P, P, P, P, P, P, I, I, P],
    "scnerf_ray_reduce": [P, P, P, P, P, I, I, I, I, P],
    "scnerf_gather_f32": [P, P, P, LL, P],
    "scnerf_mlp_layout_info": [I, P, I],
    "scnerf_mlp_fwd": [I, P, P, I, I, P, P, P, LL, P],
    "scnerf_coarse_stage_fwd": [P, I, P, P, I, P, P, P, I, P, P, P, P, P, P, P, P, I, I, P],
    "scnerf_mlp_bwd": [I, P, P, P, I, I, P, P, P, P, P, LL, P],
    "scnerf_nerf_param_count": [I],
    "scnerf_nerf_wgrad": [I, P, P, P, LL, I, P, P, I, P],
    "scnerf_wgrad": [P, I, I, I, I, P, I, I, I, I, LL, I, P, P, I, I, P, P],
    "scnerf_vecmat": [P, P, I, LL, I, P, P, P, P],
    "scnerf_wgrad_arithmetic": [I],
    "scnerf_wgrad256_chunks": [I],
    "scnerf_h3_pack": [P, P, P, P, LL, P, P, LL, P, P, P, P],
    "scnerf_mlp_fwd_h3": [I, P, P, I, I, P, P, P, P, P, LL, P, I, LL, P],
    "scnerf_mlp_bwd_h3": [I, P, P, P, I, I, P, P, P, P, P, P, P, LL, P, I, LL, P],
    "scnerf_coarse_stage_fwd_h3": [P, I, P, P, I, P, P, P, P, P, I, P, P, P, P, P, P, P, P, I, I, P, I, LL, P],
    "scnerf_fine_stage_fwd_h3": [P, I, P, P, P, I, P, P, P, P, P, I, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, P, I, LL, P],
    "scnerf_nerf_wgrad_h3": [I, P, P, P, LL, I, P, P, I, P, P, P, P, P, P],
    "scnerf_wgrad256_half": [P, P, LL, I, P, P, P, P, P, P],
    "scnerf_wgrad_half_narrow": [P, I, P, I, I, I, LL, I, P, P, P, P, P, I, LL, P],
}


# functions returning long long instead of a status
SIZE_FUNCS = {"scnerf_mlp_save_floats": [I, LL], "scnerf_mlp_grad_floats": [LL],
              "scnerf_wgrad_workspace_floats": [I, I, I],
              "scnerf_nerf_wgrad_workspace_floats": [I],
              "scnerf_h3_scale_floats": [],
              "scnerf_wgrad_chunk_samples": [LL, I],
              "scnerf_camera_bwd_workspace_floats": [I]}


def bind(lib: ctypes.CDLL) -> ctypes.CDLL:
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError == a declared symbol is missing
        fn.argtypes = args
        fn.restype = c_int
    for name, args in SIZE_FUNCS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_longlong
    return lib


_lib = None


class ScnerfLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        # torch first: the process must end up with ONE HIP runtime (torch's), so that the streams and
        # device pointers handed over by PyTorch mean the same thing inside libscnerf_hip.so
        import torch  # noqa: F401
        if not os.path.isfile(LIB_PATH):
            raise ScnerfLibraryError(
                "libscnerf_hip.so is not built (%s). Run `python -m scnerf_amd.csrc.build` "
                "(or __graft_entry__.build()); scnerf_amd has no CPU fallback." % LIB_PATH)
        _lib = bind(ctypes.CDLL(LIB_PATH))
        if _lib.scnerf_abi_version() != 4:
            raise ScnerfLibraryError("ABI version mismatch")
    return _lib


def on_device(t) -> bool:
    """True when tensor `t` lives in the memory the loaded library computes on (the GPU's HBM).  Every
    residency check of the host layer goes through here."""
    return bool(t.is_cuda)


def current_stream():
    """The raw hipStream_t of torch's current stream (kernels are enqueued there)."""
    import torch
    return torch.cuda.current_stream().cuda_stream


def check(status: int, what: str):
    if status != 0:
        raise RuntimeError("%s failed with status %d" % (what, status))
