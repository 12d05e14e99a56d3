"""Mirror of the reference's learnable camera models (/root/reference model/camera_model.py):
same class names, constructor arguments, parameter names / shapes / requires_grad flags and public
methods, so `run_nerf.py` (hasattr probes, requires_grad_ curriculum toggles, checkpoints) works
unchanged.  Ray generation itself does not go through these methods: `get_rays_*` hands the raw
parameter tensors to the fused HIP kernel.

Star-import surface.  `run_nerf.py` obtains `torch`, `np`, `nn`, `wandb`, `Image`, `sys` and every helper
of camera_utils only through `from model.camera_model import *` (/root/reference NeRF/run_nerf.py:58,
model/camera_model.py:1-9), so this module re-exports exactly those names (`__all__` below).  Where the
`wandb` package is not installed an offline stand-in with the three entry points the scripts use
(`init`, `log`, `Image`) takes its place; it keeps what was logged in `wandb.history`."""
from __future__ import annotations

import sys
import types

import numpy as np
import torch
import torch.nn as nn
from PIL import Image

from .camera_functional import CameraMatricesFunction, MatricesMemo, UpsampleGridFunction
from .camera_utils import *                                     # noqa: F401,F403  (re-exported, as the reference does)
from .camera_utils import __all__ as _camera_utils_all
from .camera_utils import (get_44_rotation_matrix_from_33_rotation_matrix, intrinsic_param_to_K,
                           ortho2rotation, rotation2orth, to_pil_normalize)


def _offline_wandb():
    """Stand-in for the wandb package (logging is out of scope; the training script must still run)."""
    m = types.ModuleType("wandb")
    m.__doc__ = "offline stand-in created by scnerf_amd.camera_model (wandb is not installed)"
    m.history, m.run = [], None

    class _Image:
        def __init__(self, data, caption=None):
            self.image, self.caption = data, caption

    def init(*args, **kwargs):
        m.run = types.SimpleNamespace(name=kwargs.get("name"), project=kwargs.get("project"))
        return m.run

    def log(data, step=None, **kwargs):
        m.history.append((step, dict(data)))
        del m.history[:-64]                                        # bounded: the loop logs every iteration

    m.Image, m.init, m.log, m.finish = _Image, init, log, lambda *a, **k: None
    return m


try:
    import wandb
except ImportError:
    wandb = _offline_wandb()

__all__ = sorted(set(_camera_utils_all) | {
    "torch", "nn", "np", "Image", "wandb", "sys", "CameraModel", "PinholeModelRotNoiseLearning10kRayoRayd",
    "PinholeModelRotNoiseLearning10kRayoRaydDistortion"})


class CameraModel(nn.Module):
    """Base class (reference :12-52)."""

    def __init__(self, intrinsics, extrinsics, args, H, W):
        nn.Module.__init__(self)
        self.args = args
        self.H, self.W = H, W
        self.model_name = args.camera_model
        self.ray_o_noise_scale = args.ray_o_noise_scale
        self.ray_d_noise_scale = args.ray_d_noise_scale
        self.extrinsics_noise_scale = args.extrinsics_noise_scale
        self.intrinsics_noise_scale = args.intrinsics_noise_scale

    def get_ray_d_noise(self):
        """[H*W, 3]: the direction-noise grid upsampled bilinearly to the image, times its scale (:24-34)."""
        return UpsampleGridFunction.apply(self.ray_d_noise, self.ray_d_noise_scale, self.H, self.W)

    def get_ray_o_noise(self):
        return UpsampleGridFunction.apply(self.ray_o_noise, self.ray_o_noise_scale, self.H, self.W)

    def get_extrinsic(self):
        raise Exception("function get_intrinsic not implemented!")

    def get_intrinsic(self):
        raise Exception("function get_extrinsic not implemented!")

    def log_noises(self, gt_intrinsic, gt_extrinsic):
        """(scalars, images) for the logger, keys and values as the reference (:54-117): the current
        intrinsics and their absolute errors against `gt_intrinsic`, extrinsic statistics and mean absolute
        error against `gt_extrinsic`, statistics + a normalised picture of each upsampled ray-noise field, and
        the distortion coefficients where the model has them.  Values stay 0-dim tensors like there
        (wandb accepts them); `run_nerf.py:607-613` unpacks the pair.  (The reference stores the *mean* under
        "camera/intrinsic_noise_std" as well, :58-61 -- reproduced.)"""
        scalars, images = {}, {}
        K = self.get_intrinsic()
        scalars["camera/intrinsic_noise_mean"] = K.abs().mean()
        scalars["camera/intrinsic_noise_std"] = K.abs().mean()
        for name, (r, c) in (("fx", (0, 0)), ("fy", (1, 1)), ("cx", (0, 2)), ("cy", (1, 2))):
            scalars["camera/" + name] = K[r][c]
        for name, (r, c) in (("fx", (0, 0)), ("fy", (1, 1)), ("cx", (0, 2)), ("cy", (1, 2))):
            scalars["camera/%s_err" % name] = (K[r][c] - gt_intrinsic[r][c]).abs()
        if hasattr(self, "extrinsics_noise"):
            E = self.get_extrinsic()
            scalars["camera/extrinsic_noise_mean"] = E.abs().mean()
            scalars["camera/extrinsic_noise_std"] = E.abs().std()
            scalars["camera/extrinisic_err"] = (E - gt_extrinsic).abs().mean()      # (sic) the reference's key
        for tag, getter in (("ray_o_noise", self.get_ray_o_noise), ("ray_d_noise", self.get_ray_d_noise)):
            if hasattr(self, tag):
                field = getter()
                scalars["camera/%s_mean" % tag] = field.abs().mean()
                scalars["camera/%s_std" % tag] = field.abs().std()
                images["camera/" + tag] = to_pil_normalize(field.reshape(self.H, self.W, 3))
        if hasattr(self, "distortion_noise"):
            k1, k2 = self.get_distortion()
            scalars["camera/k1"], scalars["camera/k2"] = k1, k2
        return scalars, images


class _PinholeRotNoise(CameraModel):
    # False (the default) = the reference's structure: every getter call builds a graph of its own, so a script may
    # backpropagate one loss through K and, separately, another through E.  True: get_intrinsic() and get_extrinsic() of one
    # parameter version share ONE autograd node (one launch each way for the pair instead of one per call) -- for loops that
    # run ONE backward per step, as run_nerf.py's train() does: scnerf_amd.dropin.install() and bench.py opt in
    # (per instance or on the class).
    share_matrix_node = False

    def focal_xy(self):
        """[fx, fy] of get_intrinsic() without building the 4x4 matrix (what the NDC warp needs per step)."""
        p = self.intrinsics_initial[:2]
        r = self.intrinsics_noise[:2] * self.intrinsics_noise_scale
        return p + (r * p if self.multiplicative_noise else r)

    def _matrices(self):
        """(K, E) of the parameters where they live on the device: ONE launch each way for the pair
        (CameraMatricesFunction) instead of the ~35 tensor ops below and the ~70 of their backward -- most of what one
        projected-ray-distance term used to launch.  get_intrinsic() and get_extrinsic() of the same parameter values
        share the node: the pair is memoised per version of the four tensors (and per grad mode) until the node's
        backward has run.  CPU parameters (construction, host-side logging) take the tensor ops."""
        from . import _capi
        if not _capi.on_device(self.intrinsics_noise):
            return None
        if not self.share_matrix_node:
            # the reference's structure: every getter call builds a graph of its own (two separate backward passes through
            # K and through E work without retain_graph), at one launch each way per call
            return CameraMatricesFunction.apply(self.intrinsics_initial, self.intrinsics_noise, self.intrinsics_noise_scale,
                                                self.multiplicative_noise, self.extrinsics_initial, self.extrinsics_noise,
                                                self.extrinsics_noise_scale, None)
        tensors = (self.intrinsics_initial, self.intrinsics_noise, self.extrinsics_initial, self.extrinsics_noise)
        key = tuple((t.data_ptr(), t._version, bool(t.requires_grad)) for t in tensors) + (torch.is_grad_enabled(),)
        memo = self.__dict__.get("_matrices_memo")
        if memo is None:
            memo = self.__dict__["_matrices_memo"] = MatricesMemo()
        if memo.key == key:
            return memo.pair
        pair = CameraMatricesFunction.apply(self.intrinsics_initial, self.intrinsics_noise, self.intrinsics_noise_scale,
                                            self.multiplicative_noise, self.extrinsics_initial, self.extrinsics_noise,
                                            self.extrinsics_noise_scale, memo)
        memo.key, memo.pair = key, pair
        return pair

    def get_intrinsic(self):
        fused = self._matrices()
        if fused is not None:
            return fused[0]
        if self.multiplicative_noise:
            p = self.intrinsics_initial + self.intrinsics_noise * self.intrinsics_noise_scale * self.intrinsics_initial
        else:
            p = self.intrinsics_initial + self.intrinsics_noise * self.intrinsics_noise_scale
        return intrinsic_param_to_K(p)

    def get_extrinsic(self):
        fused = self._matrices()
        if fused is not None:
            return fused[1]
        e = get_44_rotation_matrix_from_33_rotation_matrix(ortho2rotation(
            self.extrinsics_initial[:, :6] + self.extrinsics_noise_scale * self.extrinsics_noise[:, :6]))
        e[..., :3, 3] = self.extrinsics_initial[:, 6:] + self.extrinsics_noise_scale * self.extrinsics_noise[:, 6:]
        return e

    def forward(self, idx):
        e = get_44_rotation_matrix_from_33_rotation_matrix(ortho2rotation(
            self.extrinsics_initial[idx, None, :6] + self.extrinsics_noise_scale * self.extrinsics_noise[idx, None, :6]))
        e[..., :3, 3] = self.extrinsics_initial[idx, 6:] + self.extrinsics_noise_scale * self.extrinsics_noise[idx, 6:]
        return self.get_intrinsic(), e.squeeze()

    def _register_common(self, intrinsics, extrinsics):
        fx, fy, tx, ty = intrinsics[0][0], intrinsics[1][1], intrinsics[0][2], intrinsics[1][2]
        extrinsics = torch.from_numpy(np.stack([np.asarray(e.cpu() if torch.is_tensor(e) else e) for e in extrinsics])).float()
        params = rotation2orth(extrinsics[:, :3, :3])
        translations = extrinsics[:, :3, 3]
        self.register_parameter("intrinsics_initial", nn.Parameter(
            torch.Tensor([float(fx), float(fy), float(tx), float(ty)]), requires_grad=False))
        self.register_parameter("extrinsics_initial", nn.Parameter(
            torch.cat([params, translations], dim=-1), requires_grad=False))


class PinholeModelRotNoiseLearning10kRayoRayd(_PinholeRotNoise):
    """Reference :120-206."""

    def __init__(self, intrinsics, extrinsics, args, H, W):
        super().__init__(intrinsics, extrinsics, args, H, W)
        self._register_common(intrinsics, extrinsics)
        ray_o_noise = torch.zeros((H // args.grid_size, W // args.grid_size, 3))
        ray_d_noise = torch.zeros((H // args.grid_size, W // args.grid_size, 3))
        self.register_parameter("intrinsics_noise", nn.Parameter(torch.zeros(4)))
        self.register_parameter("extrinsics_noise", nn.Parameter(torch.zeros_like(self.extrinsics_initial)))
        self.register_parameter("ray_o_noise", nn.Parameter(ray_o_noise))
        self.register_parameter("ray_d_noise", nn.Parameter(ray_d_noise))
        self.multiplicative_noise = args.multiplicative_noise


class PinholeModelRotNoiseLearning10kRayoRaydDistortion(_PinholeRotNoise):
    """Reference :209-312.  As there, ray_o_noise and ray_d_noise wrap ONE tensor (:224, :257-262):
    two Parameters (two autograd leaves) over the same storage until .to()/.cuda() copies them."""

    def __init__(self, intrinsics, extrinsics, args, H, W, k=None):
        super().__init__(intrinsics, extrinsics, args, H, W)
        self._register_common(intrinsics, extrinsics)
        ray_noise = torch.zeros((H // args.grid_size, W // args.grid_size, 3))
        if k is not None:
            self.register_parameter("distortion_initial", nn.Parameter(torch.tensor([k[0], k[1]]), requires_grad=False))
        else:
            self.register_parameter("distortion_initial", nn.Parameter(torch.zeros(2), requires_grad=False))
        self.register_parameter("intrinsics_noise", nn.Parameter(torch.zeros(4)))
        self.register_parameter("extrinsics_noise", nn.Parameter(torch.zeros_like(self.extrinsics_initial)))
        self.register_parameter("ray_o_noise", nn.Parameter(ray_noise))
        self.register_parameter("ray_d_noise", nn.Parameter(ray_noise))
        self.register_parameter("distortion_noise", nn.Parameter(torch.zeros(2)))
        self.multiplicative_noise = args.multiplicative_noise if hasattr(args, "multiplicative_noise") else False

    def get_distortion(self):
        return self.distortion_initial + self.distortion_noise * self.args.distortion_noise_scale
