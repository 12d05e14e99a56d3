"""Makes the reference's own module names resolve to this package, so that an unmodified
`NeRF/run_nerf.py` (`from render import render, render_path`, `from get_rays import ...`,
`from run_nerf_helpers import ...`, `from create_nerf import create_nerf`,
`from model.camera_model import *`, `from model.ray_dist_loss import ...`:
/root/reference NeRF/run_nerf.py:13-62) runs on the HIP path.

    import scnerf_amd.dropin; scnerf_amd.dropin.install()     # before run_nerf.py's imports
"""
import importlib
import sys
import types

_MAP = {
    "render": "scnerf_amd.render",
    "get_rays": "scnerf_amd.get_rays",
    "run_nerf_helpers": "scnerf_amd.run_nerf_helpers",
    "create_nerf": "scnerf_amd.create_nerf",
    "camera_dict": "scnerf_amd.camera_dict",
    "camera_model": "scnerf_amd.camera_model",
    "model.camera_model": "scnerf_amd.camera_model",
    "model.camera_utils": "scnerf_amd.camera_utils",
    "model.ray_dist_loss": "scnerf_amd.ray_dist_loss",
}


# the NeRF++ scripts import these as top-level modules of nerfplusplus/ (ddp_train_nerf.py:8-25)
_MAP_NERFPP = {
    "nerf_network": "scnerf_amd.nerfplusplus.nerf_network",
    "ddp_model": "scnerf_amd.nerfplusplus.ddp_model",
    "nerf_sample_ray_split": "scnerf_amd.nerfplusplus.nerf_sample_ray_split",
    "camera_model": "scnerf_amd.camera_model",
    "model.camera_model": "scnerf_amd.camera_model",
    "model.camera_utils": "scnerf_amd.camera_utils",
    "model.ray_dist_loss": "scnerf_amd.ray_dist_loss",
}


def install_nerfplusplus():
    """As install(), for the module names nerfplusplus/ddp_train_nerf.py imports; its own per-ray helpers
    (intersect_sphere, perturb_samples, sample_pdf) live in scnerf_amd.nerfplusplus.ddp_train_nerf and
    `create_nerf` in scnerf_amd.nerfplusplus.create_nerf (same names / signatures).  The reference's
    render_ray_from_camera is only one function of nerf_sample_ray_split.py: the remaining names of that
    module (RaySamplerSingleImage ...) are host-side data handling and stay the reference's."""
    for alias, target in _MAP_NERFPP.items():
        if alias == "nerf_sample_ray_split":
            continue                      # patched function-wise below when the reference module is importable
        sys.modules[alias] = importlib.import_module(target)
    if "model" not in sys.modules or not hasattr(sys.modules["model"], "__path__"):
        pkg = types.ModuleType("model")
        pkg.__path__ = []
        sys.modules["model"] = pkg
    for name in ("camera_model", "camera_utils", "ray_dist_loss"):
        setattr(sys.modules["model"], name, sys.modules["model." + name])
    try:
        ref = importlib.import_module("nerf_sample_ray_split")
        ref.render_ray_from_camera = importlib.import_module(_MAP_NERFPP["nerf_sample_ray_split"]).render_ray_from_camera
    except Exception:
        sys.modules["nerf_sample_ray_split"] = importlib.import_module(_MAP_NERFPP["nerf_sample_ray_split"])
    return sorted(_MAP_NERFPP)


def install():
    for alias, target in _MAP.items():
        sys.modules[alias] = importlib.import_module(target)
    if "model" not in sys.modules or not hasattr(sys.modules["model"], "__path__"):
        pkg = types.ModuleType("model")
        pkg.__path__ = []
        sys.modules["model"] = pkg
    sys.modules["model"].camera_model = sys.modules["model.camera_model"]
    sys.modules["model"].camera_utils = sys.modules["model.camera_utils"]
    sys.modules["model"].ray_dist_loss = sys.modules["model.ray_dist_loss"]
    return sorted(_MAP)
