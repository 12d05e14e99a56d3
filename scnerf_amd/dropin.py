"""Makes the reference's own module names resolve to this package, so that an unmodified
`NeRF/run_nerf.py` (`from render import render, render_path`, `from get_rays import ...`,
`from run_nerf_helpers import ...`, `from create_nerf import create_nerf`,
`from model.camera_model import *`, `from model.ray_dist_loss import ...`:
/root/reference NeRF/run_nerf.py:13-62) runs on the HIP path.

    import scnerf_amd.dropin; scnerf_amd.dropin.install()     # before run_nerf.py's imports
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

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
    # model/prd_evaluation.py:2 imports it as a top-level module (load_llff.py:6 puts ../model on sys.path)
    "ray_dist_loss": "scnerf_amd.ray_dist_loss",
    # run_nerf.py:73 `from prd_evaluation import projected_ray_distance_evaluation` (model/ is on sys.path)
    "prd_evaluation": "scnerf_amd.prd_evaluation",
    "model.prd_evaluation": "scnerf_amd.prd_evaluation",
}


# the NeRF++ scripts import these as top-level modules of nerfplusplus/ (ddp_train_nerf.py:8-25)
_MAP_NERFPP = {
    # nerfplusplus/ddp_train_nerf.py:16 `from create_nerf import create_nerf`: the reference's wraps the networks in
    # DistributedDataParallel and builds torch optimizers; the mirror builds the fused optimizer and attaches ONE
    # all-reduce of its gradient arena to step() instead (same signature and return values)
    "create_nerf": "scnerf_amd.nerfplusplus.create_nerf",
    "nerf_network": "scnerf_amd.nerfplusplus.nerf_network",
    "ddp_model": "scnerf_amd.nerfplusplus.ddp_model",
    "nerf_sample_ray_split": "scnerf_amd.nerfplusplus.nerf_sample_ray_split",
    "camera_model": "scnerf_amd.camera_model",
    "model.camera_model": "scnerf_amd.camera_model",
    "model.camera_utils": "scnerf_amd.camera_utils",
    "model.ray_dist_loss": "scnerf_amd.ray_dist_loss",
}


class _ModelPackageFallback(importlib.abc.MetaPathFinder):
    """LAST meta-path entry: if no real `model` package is importable (the reference tree is not on
    sys.path), `import model.<mirrored module>` still has to succeed, so an empty package named `model`
    is synthesised.  When the reference's model/ directory IS importable it wins, and its un-mirrored
    modules (`model.reprojection`, `model.lookup`: matchers, host-side) keep resolving from there while the
    mirrored ones come out of sys.modules."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname != "model":
            return None
        spec = importlib.machinery.ModuleSpec("model", None, is_package=True)
        spec.submodule_search_locations = []
        return spec


def _register(mapping):
    for alias, target in mapping.items():
        sys.modules[alias] = importlib.import_module(target)
    stale = sys.modules.get("model")
    if stale is not None and not getattr(stale, "__file__", None) and not list(getattr(stale, "__path__", [])):
        del sys.modules["model"]                    # an empty stand-in from an earlier install(): re-resolve
    if not any(isinstance(f, _ModelPackageFallback) for f in sys.meta_path):
        sys.meta_path.append(_ModelPackageFallback())
    return sorted(mapping)


def install_nerfplusplus():
    """As install(), for the module names nerfplusplus/ddp_train_nerf.py imports; its own per-ray helpers
    (intersect_sphere, perturb_samples, sample_pdf) live in scnerf_amd.nerfplusplus.ddp_train_nerf and
    `create_nerf` in scnerf_amd.nerfplusplus.create_nerf (same names / signatures).  The reference's
    render_ray_from_camera is only one function of nerf_sample_ray_split.py: the remaining names of that
    module (RaySamplerSingleImage ...) are host-side data handling and stay the reference's."""
    names = _register({k: v for k, v in _MAP_NERFPP.items() if k != "nerf_sample_ray_split"})
    try:
        ref = importlib.import_module("nerf_sample_ray_split")    # patched function-wise when importable
        ref.render_ray_from_camera = importlib.import_module(_MAP_NERFPP["nerf_sample_ray_split"]).render_ray_from_camera
    except Exception:
        sys.modules["nerf_sample_ray_split"] = importlib.import_module(_MAP_NERFPP["nerf_sample_ray_split"])
    return sorted(names + ["nerf_sample_ray_split"])


def install(share_matrix_node=True):
    """Registers the mirrors under the reference's module names (see the module docstring).  run_nerf.py's train() runs ONE
    backward per step (NeRF/run_nerf.py:598), so the camera model's K / E pair may share one autograd node there
    (`share_matrix_node`: camera_model._PinholeRotNoise); pass False for a script that backpropagates through K and E
    separately."""
    names = _register(_MAP)
    from . import camera_model
    camera_model._PinholeRotNoise.share_matrix_node = bool(share_matrix_node)
    return names
