"""scnerf_amd -- MI355X-native (gfx950) implementation of SCNeRF's per-ray render path and
differentiable camera ray generator behind the reference's own Python surface.

Module names mirror the reference's (`render`, `run_nerf_helpers`, `create_nerf`, `get_rays`,
`camera_model`, `camera_dict`); `scnerf_amd.dropin.install()` puts them on `sys.modules` under
those top-level names so that `NeRF/run_nerf.py` imports them unchanged.

All arithmetic runs in hand-written HIP kernels (scnerf_amd/csrc) loaded through the C ABI of
include/scnerf_hip.h; there is no CPU or eager-PyTorch fallback."""

__version__ = "0.1.0"
