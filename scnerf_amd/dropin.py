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
