"""render_ray_from_camera (nerfplusplus/nerf_sample_ray_split.py:196-257) on the camera kernels."""
from __future__ import annotations

import numpy as np
import torch

from .. import _capi
from ..ops import _cam_common, _p, _stream


def _cf(t):
    return None if t is None else t.detach().contiguous().float()


class _NppCameraRays(torch.autograd.Function):
    """apply(meta, select, dist, extrinsic, intr_noise, extr_noise, grid_o, grid_d) -> rays_o, rays_d"""

    @staticmethod
    def forward(ctx, meta, select, dist, extrinsic, intr_noise, extr_noise, grid_o, grid_d):
        cam = dict(meta)
        cam.update(extrinsic=_cf(extrinsic), intr_noise=_cf(intr_noise), extr_noise=_cf(extr_noise),
                   grid_o=_cf(grid_o), grid_d=_cf(grid_d))
        sel = select.detach().contiguous().long()
        dist_c = _cf(dist)
        n = sel.numel()
        dev = cam["intr_init"].device
        ro = torch.empty((n, 3), dtype=torch.float32, device=dev)
        rd = torch.empty((n, 3), dtype=torch.float32, device=dev)
        common = _cam_common(cam)          # (kps, cam_idx, single, ext, n_ext, intr..., H, W)
        st = _capi.load().scnerf_npp_camera_rays_fwd(_p(sel), _p(dist_c), common[2], common[3], *common[5:], _p(ro),
                                                     _p(rd), n, _stream())
        _capi.check(st, "scnerf_npp_camera_rays_fwd")
        ctx.state = (cam, sel, dist_c, n)
        return ro, rd

    @staticmethod
    def backward(ctx, g_o, g_d):
        cam, sel, dist_c, n = ctx.state
        lib = _capi.load()
        dev = cam["intr_init"].device
        C = int(cam["extr_init"].shape[0])
        ext = cam.get("extrinsic")
        d_in = torch.empty(4, dtype=torch.float32, device=dev)
        d_ex = torch.empty((C, 9), dtype=torch.float32, device=dev) if ext is None else None
        d_go = torch.empty_like(cam["grid_o"]) if cam.get("grid_o") is not None else None
        d_gd = torch.empty_like(cam["grid_d"]) if cam.get("grid_d") is not None else None
        d_E = torch.empty((1, 4, 4), dtype=torch.float32, device=dev) if ext is not None else None
        d_dist = torch.empty(2, dtype=torch.float32, device=dev) if dist_c is not None else None
        ws = torch.empty(lib.scnerf_camera_bwd_workspace_floats(max(C, 1)), dtype=torch.float32, device=dev)
        common = _cam_common(cam)
        st = lib.scnerf_npp_camera_rays_bwd(_p(sel), _p(dist_c), common[2], common[3], *common[5:], _p(_cf(g_o)),
                                            _p(_cf(g_d)), _p(d_in), _p(d_ex), _p(d_go), _p(d_gd), _p(d_E), _p(d_dist),
                                            _p(ws), n, _stream())
        _capi.check(st, "scnerf_npp_camera_rays_bwd")
        need = ctx.needs_input_grad
        return (None, None, d_dist if need[2] else None, (d_E[0] if d_E is not None else None) if need[3] else None,
                d_in if need[4] else None, d_ex if (need[5] and d_ex is not None) else None,
                d_go if need[6] else None, d_gd if need[7] else None)


def render_ray_from_camera(camera_model, camera_idx, select_inds, rank, extrinsic=None):
    """Rays through the centres of the pixels `select_inds` (row-major indices into the H x W image) of
    camera `camera_idx` of the learnable camera model -- or of the pose `extrinsic` (numpy [4,4]) when
    camera_idx is None -- with radial distortion if the model has it; returns (rays_o, rays_d, depth)
    exactly like the reference (:196-257; `depth` is c2w.T[2, 3] broadcast, :256)."""
    W, H = camera_model.W, camera_model.H
    dev = camera_model.intrinsics_initial.device
    if not _capi.on_device(camera_model.intrinsics_initial):
        raise RuntimeError("the camera model must be on the GPU (scnerf_amd has no CPU path)")
    ext = None
    if camera_idx is None:
        assert extrinsic is not None
        ext = torch.as_tensor(np.asarray(extrinsic), dtype=torch.float32).to(dev)
    sel = torch.as_tensor(select_inds).to(dev).reshape(-1).long()
    has_o, has_d = hasattr(camera_model, "ray_o_noise"), hasattr(camera_model, "ray_d_noise")
    meta = dict(single_idx=0 if camera_idx is None else int(camera_idx),
                intr_init=camera_model.intrinsics_initial.detach().contiguous().float(),
                intr_scale=float(camera_model.intrinsics_noise_scale),
                multiplicative=bool(getattr(camera_model, "multiplicative_noise", False)),
                extr_init=camera_model.extrinsics_initial.detach().contiguous().float(),
                extr_scale=float(camera_model.extrinsics_noise_scale),
                scale_o=float(camera_model.ray_o_noise_scale), scale_d=float(camera_model.ray_d_noise_scale),
                H=int(H), W=int(W), n=int(sel.numel()))
    dist = camera_model.get_distortion() if hasattr(camera_model, "distortion_noise") else None
    rays_o, rays_d = _NppCameraRays.apply(meta, sel, dist, ext, camera_model.intrinsics_noise,
                                          camera_model.extrinsics_noise,
                                          camera_model.ray_o_noise if has_o else None,
                                          camera_model.ray_d_noise if has_d else None)
    # c2w.T[2, 3] = c2w[3, 2]: the bottom row of a rigid transform -- a constant the reference broadcasts
    c2w_32 = ext[3, 2] if ext is not None else torch.zeros((), device=dev)
    depth = c2w_32 * torch.ones((rays_o.shape[0],), device=dev)
    return rays_o, rays_d, depth
