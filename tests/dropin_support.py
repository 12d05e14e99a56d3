"""TEST INFRASTRUCTURE for the drop-in tests (tests/test_dropin_run_nerf.py).

* `install_stubs()` puts minimal stand-ins on sys.modules for the packages the reference's training
  script imports that are neither on the path being replaced nor installed here (imageio, cv2,
  configargparse, piqa, torchvision, the un-vendored `thirdparty.*` submodules).  None of them computes
  anything on the render path.
* `import_reference_run_nerf()` imports the UNMODIFIED /root/reference/NeRF/run_nerf.py with
  `scnerf_amd.dropin.install()` active.  The reference tree exists only in the build container; callers skip
  when it is absent.
* `synthetic_llff()` is the stand-in for `load_llff_data` (same return structure, load_llff.py:241-379).
* `train_argv()` builds a command line for `config_parser()`.
"""
import argparse
import contextlib
import importlib
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = os.environ.get("SCNERF_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REF_ROOT, "NeRF", "run_nerf.py"))


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__stub__ = True
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent:
        if parent not in sys.modules:
            _module(parent)
        setattr(sys.modules[parent], leaf, m)
        if not hasattr(sys.modules[parent], "__path__"):
            sys.modules[parent].__path__ = []
    return m


class _ConfigArgumentParser(argparse.ArgumentParser):
    """configargparse.ArgumentParser without config-file support (`is_config_file` options are plain)."""

    def add_argument(self, *names, **kw):
        kw.pop("is_config_file", None)
        return super().add_argument(*names, **kw)


class _ZeroMetric(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, x, y):
        return torch.zeros((), device=x.device)


def _identity_alignment(traj_a, traj_b, traj_c=None):
    """thirdparty/nerfmm align_ate_c2b_use_a2b: here the poses to transform come back unchanged."""
    out = traj_c if traj_c is not None else traj_a
    return out.clone().float()


def install_stubs(force=()):
    """Returns the names that were stubbed (already importable packages are left alone unless in `force`)."""
    made = []

    def need(name):
        if name in force:
            return True
        if name in sys.modules:
            return False
        try:
            importlib.import_module(name)
            return False
        except Exception:
            return True

    saved = []
    if need("imageio"):
        _module("imageio", imwrite=lambda *a, **k: None, mimwrite=lambda *a, **k: None,
                imread=lambda *a, **k: saved)
        made.append("imageio")
    if need("cv2"):
        _module("cv2", SIFT_create=lambda *a, **k: types.SimpleNamespace(detectAndCompute=None))
        made.append("cv2")
    if need("configargparse"):
        _module("configargparse", ArgumentParser=_ConfigArgumentParser)
        made.append("configargparse")
    if need("piqa"):
        _module("piqa")
        _module("piqa.ssim", SSIM=_ZeroMetric)
        _module("piqa.lpips", LPIPS=_ZeroMetric)
        made.append("piqa")
    if need("wandb"):                       # scripts that `import wandb` themselves get the package's offline stand-in
        from scnerf_amd.camera_model import wandb as offline
        sys.modules["wandb"] = offline
        made.append("wandb")
    if need("torchvision"):
        _module("torchvision")
        _module("torchvision.transforms", ToPILImage=lambda: (lambda t: t))
        made.append("torchvision")
    # un-vendored submodules of the reference (empty directories in /root/reference/thirdparty)
    for name in ("thirdparty", "thirdparty.ATE", "thirdparty.superglue", "thirdparty.superglue.models",
                 "thirdparty.nerfmm", "thirdparty.nerfmm.utils"):
        _module(name)
    _module("thirdparty.superglue.models.matching", Matching=type("Matching", (torch.nn.Module,), {}))
    _module("thirdparty.nerfmm.utils.align_traj", align_ate_c2b_use_a2b=_identity_alignment)
    made.append("thirdparty")
    return made


@contextlib.contextmanager
def _cwd(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


_REF_MODULE_NAMES = ("run_nerf", "config_argparse", "load_llff", "load_blender", "prd_evaluation", "reprojection",
                     "lookup", "src", "src.utils", "unit_tests", "unit_tests.noise_injection_test",
                     "unit_tests.visualize_matches", "unit_tests.utils", "model", "model.reprojection",
                     "model.lookup", "model.prd_evaluation")


def import_reference_run_nerf():
    """-> the module object of the unmodified NeRF/run_nerf.py, imported the way `python run_nerf.py` sees
    the world: cwd = NeRF/ (its sys.path entries are relative), scnerf_amd.dropin.install() done first."""
    assert reference_available()
    import scnerf_amd.dropin as dropin
    for n in _REF_MODULE_NAMES:
        sys.modules.pop(n, None)
    # oracle/ref_import.py (golden-vector pinning) may have put the REFERENCE's own hot-path modules under
    # these top-level names earlier in the same test process; install() overwrites them with the mirrors.
    install_stubs()
    dropin.install()
    nerf_dir = os.path.join(REF_ROOT, "NeRF")
    saved_path = list(sys.path)
    sys.path.insert(0, nerf_dir)
    try:
        with _cwd(nerf_dir):
            anomaly = torch.is_anomaly_enabled()
            mod = importlib.import_module("run_nerf")
            torch.autograd.set_detect_anomaly(anomaly)
    finally:
        # keep what the script itself inserted (relative entries are harmless outside NeRF/) out of the way
        sys.path[:] = saved_path
    return mod


def forget_reference_modules():
    for n in _REF_MODULE_NAMES:
        sys.modules.pop(n, None)


# ---- synthetic LLFF-shaped scene --------------------------------------------------------------------

def _look_at_pose(center, target=(0.0, 0.0, -4.0)):
    """camera-to-world [3,4] in the LLFF / OpenGL convention (-z forward, y up)."""
    c = np.asarray(center, np.float64)
    z = c - np.asarray(target, np.float64)
    z /= np.linalg.norm(z)
    x = np.cross([0.0, 1.0, 0.0], z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.concatenate([np.stack([x, y, z], 1), c[:, None]], 1)


def synthetic_llff(H=24, W=32, n_images=5, focal=30.0, seed=0, device="cpu"):
    """(images, poses [N,3,5], bds [N,2], render_poses, i_test, (gt_intrinsic, gt_extrinsic)) with the types
    load_llff_data returns: numpy float32 images / poses / bounds, a torch render path, ground truth as
    tensors on `device` (the reference puts them on the GPU)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
    images, poses = [], []
    for i in range(n_images):
        ph = 2 * np.pi * i / n_images
        img = np.stack([0.5 + 0.4 * np.sin(6 * xx + ph), 0.5 + 0.4 * np.cos(5 * yy - ph),
                        0.5 + 0.3 * np.sin(4 * (xx + yy) + ph)], -1)
        images.append(np.clip(img + 0.02 * rng.randn(H, W, 3), 0, 1))
        c2w = _look_at_pose([0.6 * np.cos(ph), 0.4 * np.sin(ph), 0.2 * np.sin(2 * ph)])
        poses.append(np.concatenate([c2w, np.array([[H], [W], [focal]], np.float64)], 1))
    images = np.stack(images).astype(np.float32)
    poses = np.stack(poses).astype(np.float32)
    bds = np.tile(np.array([[1.2, 8.0]], np.float32), (n_images, 1))
    render_poses = torch.from_numpy(poses[:2].copy())
    i_test = np.array([0])
    gt_intrinsic = torch.tensor([[focal, 0, W // 2, 0], [0, focal, H // 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
                                dtype=torch.float32, device=device)
    gt_extrinsic = torch.zeros((n_images, 4, 4), device=device)
    gt_extrinsic[:, :3, :4] = torch.from_numpy(poses[:, :, :4]).to(device)
    gt_extrinsic[:, 3, 3] = 1
    return images, poses, bds, render_poses, i_test, (gt_intrinsic, gt_extrinsic)


def train_argv(basedir, n_iters, extra=()):
    """Command line of a tiny SCNeRF run (configs/llff_data/fern style options, shrunk)."""
    return ["run_nerf.py", "--expname", "dropin", "--basedir", str(basedir), "--datadir", "synthetic/fern",
            "--dataset_type", "llff", "--factor", "8", "--llffhold", "8",
            "--N_rand", "64", "--N_samples", "8", "--N_importance", "8", "--use_viewdirs", "--raw_noise_std", "1.0",
            "--no_batching", "--N_iters", str(n_iters), "--i_print", "1",
            "--i_weights", "1000000", "--i_testset", "1000000", "--i_img", "1000000", "--i_video", "1000000",
            "--camera_model", "pinhole_rot_noise_10k_rayo_rayd", "--ray_loss_type", "none",
            "--add_ie", "2", "--add_od", "3", "--add_prd", "1000000", "--use_custom_optim",
            "--matcher", "sift"] + list(extra)


# ---- call surface: which mirrored functions a script calls, and how ----------------------------------

import ast   # noqa: E402

# functions of the mirrored modules (render, get_rays, run_nerf_helpers, create_nerf, model.ray_dist_loss)
API_FUNCTIONS = ("render", "render_path", "get_rays_kps_use_camera", "get_rays_kps_no_camera",
                 "get_rays_full_image_use_camera", "get_rays_full_image_no_camera", "get_rays_np", "create_nerf",
                 "img2mse", "mse2psnr", "fix_seeds", "preprocess_match", "proj_ray_dist_loss_single",
                 "projected_ray_distance_evaluation")
# methods of the objects create_nerf hands back (camera model, optimizer, networks)
API_METHODS = ("log_noises", "get_extrinsic", "get_intrinsic", "requires_grad_", "zero_grad", "step", "state_dict")


def call_surface(source, function=None):
    """Sorted list of [callee, n_positional, [keyword names], has_**kwargs] over every call to a mirrored
    function / method in `source` (restricted to the body of top-level function `function` if given)."""
    tree = ast.parse(source)
    if function is not None:
        tree = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == function)
    found = set()
    for node in ast.walk(tree):
        if not isinstance(node, ast.Call):
            continue
        f = node.func
        if isinstance(f, ast.Name) and f.id in API_FUNCTIONS:
            name = f.id
        elif isinstance(f, ast.Attribute) and f.attr in API_METHODS:
            name = "." + f.attr
        else:
            continue
        kws = tuple(sorted(k.arg for k in node.keywords if k.arg is not None))
        found.add((name, len(node.args), kws, any(k.arg is None for k in node.keywords)))
    return [[n, p, list(k), s] for n, p, k, s in sorted(found)]


def undefined_globals(module):
    """Names the module's code loads as globals that exist neither in the module namespace nor in
    builtins (what `from x import *` failing to provide a name would leave behind)."""
    import builtins
    import symtable
    with open(module.__file__) as f:
        src = f.read()
    missing = set()

    def visit(table):
        for sym in table.get_symbols():
            if sym.is_referenced() and (sym.is_global() or (table.get_type() == "module" and not sym.is_assigned()
                                                          and not sym.is_imported())):
                name = sym.get_name()
                if not hasattr(module, name) and not hasattr(builtins, name):
                    missing.add(name)
        for child in table.get_children():
            visit(child)
    visit(symtable.symtable(src, module.__file__, "exec"))
    return sorted(missing)


def synthetic_matcher(H, W, n=24, seed=0):
    """Stand-in for runSIFTSinglePair / runSuperGlueSinglePair (cv2 / SuperGlue): returns the structure
    preprocess_match consumes -- a list with one dict of key points [n,2] (x, y) and index pairs."""
    def match(*args, **kwargs):
        rng = np.random.RandomState(seed)
        img = next((a for a in args if torch.is_tensor(a) and a.dim() == 3), None)
        dev = img.device if img is not None else "cpu"
        kps0 = np.stack([rng.randint(2, W - 2, n), rng.randint(2, H - 2, n)], -1)
        kps1 = np.clip(kps0 + rng.randint(-1, 2, (n, 2)), 0, [W - 1, H - 1])
        pairs = np.stack([np.arange(n), np.arange(n)], -1)
        return [{"kps0": torch.from_numpy(kps0).float().to(dev), "kps1": torch.from_numpy(kps1).float().to(dev),
                 "matches": pairs}]
    return match
