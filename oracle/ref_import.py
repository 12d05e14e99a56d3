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


def _functions_from_source(path, names, namespace):
    """Executes only the named top-level function definitions of a reference source file in
    `namespace` (the file's module-level imports -- cv2, wandb, SuperGlue ... -- are not needed by them
    and are not importable here).  The code that runs is the reference's own, read from where it lies."""
    import ast
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in picked}
    if missing:
        raise RuntimeError("%s lacks %s" % (path, sorted(missing)))
    mod = ast.Module(body=picked, type_ignores=[])
    exec(compile(mod, path, "exec"), namespace)
    return namespace


def load_nerfpp():
    """Reference NeRF++ pieces (nerfplusplus/): .ddp_model (NerfNet, depth2pts_outside),
    .nerf_network (Embedder, MLPNet), .train (intersect_sphere, perturb_samples, sample_pdf of
    ddp_train_nerf.py), .rays (render_ray_from_camera of nerf_sample_ray_split.py), .camera_model."""
    if "npp" in _cache:
        return _cache["npp"]
    base = load_reference()
    root = os.path.join(REF_ROOT, "nerfplusplus")
    saved_path = list(sys.path)
    saved = {k: sys.modules.get(k) for k in ("utils", "nerf_network", "ddp_model")}
    sys.path[:0] = [root]
    try:
        for k in saved:
            sys.modules.pop(k, None)
        for name in ("cv2", "matplotlib", "matplotlib.cm", "matplotlib.backends", "matplotlib.backends.backend_agg",
                     "matplotlib.figure", "mpl_toolkits", "mpl_toolkits.axes_grid1"):   # plotting helpers of utils.py
            if name not in sys.modules:
                try:
                    __import__(name)
                except Exception:
                    sys.modules[name] = types.ModuleType(name)
        import torch
        import numpy as np
        import ddp_model as ddp_model_mod
        import nerf_network as nerf_network_mod
        import utils as utils_mod
    finally:
        sys.path[:] = saved_path
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
            else:
                sys.modules.pop(k, None)
    train_ns = {"torch": torch, "np": np, "TINY_NUMBER": utils_mod.TINY_NUMBER}
    _functions_from_source(os.path.join(root, "ddp_train_nerf.py"),
                           ["intersect_sphere", "perturb_samples", "sample_pdf"], train_ns)
    rays_ns = {"torch": torch, "np": np}
    _functions_from_source(os.path.join(root, "nerf_sample_ray_split.py"), ["render_ray_from_camera"], rays_ns)
    ns = types.SimpleNamespace(ddp_model=ddp_model_mod, nerf_network=nerf_network_mod, utils=utils_mod,
                               train=types.SimpleNamespace(**{k: train_ns[k] for k in
                                                              ("intersect_sphere", "perturb_samples", "sample_pdf")}),
                               rays=types.SimpleNamespace(render_ray_from_camera=rays_ns["render_ray_from_camera"]),
                               camera_model=base.camera_model)
    _cache["npp"] = ns
    return ns
