"""TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Imports the *unmodified* SCNeRF reference modules from /root/reference so that
golden vectors can be generated from the real implementation (the reference is a
Python code base; it cannot travel to the GPU box, so only the vectors do).

Only `oracle/gen_golden.py` and the `-m "not gpu"` pinning tests use this, and
only inside the build container where /root/reference exists.

The reference's hot-path modules import two packages that are absent here but
unused on the path (imageio in NeRF/render.py:3, wandb in model/camera_model.py:5);
they are stubbed with empty modules.  `thirdparty.ATE` (model/camera_utils.py:8)
resolves as an empty namespace package.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("SCNERF_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "NeRF", "render.py"))


_cache = {}


def load_reference():
    """Returns a namespace with the reference modules:
    .render, .helpers, .create_nerf, .get_rays, .camera_model, .camera_utils,
    .ray_dist_loss"""
    if "ns" in _cache:
        return _cache["ns"]
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    for name in ("imageio", "wandb", "tqdm"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    paths = [os.path.join(REF_ROOT, "NeRF"), REF_ROOT, os.path.join(REF_ROOT, "model")]
    saved_path = list(sys.path)
    saved_mods = {k: sys.modules.get(k) for k in
                  ("render", "run_nerf_helpers", "create_nerf", "get_rays",
                   "camera_model", "camera_dict", "model", "model.camera_utils",
                   "model.camera_model")}
    # our own package mirrors some of these module names *inside* scnerf_amd/, never
    # at top level, so there is no clash; still isolate sys.path while importing.
    sys.path[:0] = paths
    try:
        import torch
        anomaly = torch.is_anomaly_enabled()
        import run_nerf_helpers as helpers      # NeRF/run_nerf_helpers.py (turns anomaly mode on, :7)
        torch.autograd.set_detect_anomaly(anomaly)
        import get_rays as get_rays_mod          # NeRF/get_rays.py
        import render as render_mod              # NeRF/render.py
        import camera_model as camera_model_mod  # model/camera_model.py
        import create_nerf as create_nerf_mod    # NeRF/create_nerf.py
        from model import camera_utils as camera_utils_mod
        from model import ray_dist_loss as ray_dist_loss_mod   # model/ray_dist_loss.py
    finally:
        sys.path[:] = saved_path
    ns = types.SimpleNamespace(
        render=render_mod, helpers=helpers, create_nerf=create_nerf_mod,
        get_rays=get_rays_mod, camera_model=camera_model_mod,
        camera_utils=camera_utils_mod, ray_dist_loss=ray_dist_loss_mod)
    _cache["ns"] = ns
    return ns
