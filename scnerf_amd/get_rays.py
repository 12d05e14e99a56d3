"""Mirror of the reference's `get_rays` module (/root/reference NeRF/get_rays.py) plus the two NDC
warps of NeRF/render.py:357-396; the arithmetic runs in the HIP ray-generator kernels
(scnerf_amd/csrc/camera_rays.hip) through scnerf_amd.camera_functional."""
from __future__ import annotations

import numpy as np


def _cf():
    from . import camera_functional
    return camera_functional


def get_rays_full_image_no_camera(H, W, focal, extrinsic):
    """Pinhole rays of every pixel of one image (reference :5-23)."""
    assert extrinsic.dim() == 2
    return _cf().pinhole_rays(H, W, focal, extrinsic, None)


def get_rays_kps_no_camera(H, W, focal, extrinsic, kps_list):
    """Pinhole rays at integer pixel coordinates kps_list [N, >=2] (x, y, ...) (reference :75-90)."""
    assert kps_list[:, 0].max() < W
    assert kps_list[:, 1].max() < H
    assert extrinsic.dim() == 2
    return _cf().pinhole_rays(H, W, focal, extrinsic, kps_list)


def get_rays_full_image_use_camera(H, W, camera_model, idx_in_camera_param=None, extrinsic=None):
    """Rays of every pixel through the learnable camera model (reference :26-72)."""
    return _cf().camera_rays(H, W, camera_model, None, idx_in_camera_param, extrinsic)


def get_rays_kps_use_camera(H, W, camera_model, kps_list, idx_in_camera_param=None, extrinsic=None):
    """Rays at key points kps_list [N,2] (x, y) through the learnable camera model (reference :93-148)."""
    assert kps_list[:, 0].max() < W
    assert kps_list[:, 1].max() < H
    assert (idx_in_camera_param is None and not extrinsic is None or
            not idx_in_camera_param is None and extrinsic is None)
    return _cf().camera_rays(H, W, camera_model, kps_list, idx_in_camera_param, extrinsic)


def get_rays_np(H, W, focal, extrinsic):
    """Host-side numpy variant used to pre-bake rays when there is no camera model (reference
    :151-165).  Pure numpy by definition of the API (it returns numpy arrays)."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - W * .5) / focal, -(j - H * .5) / focal, -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * extrinsic[:3, :3], -1)
    rays_o = np.broadcast_to(extrinsic[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """NeRF/render.py:357-374."""
    return _cf().ndc(H, W, focal, focal, near, rays_o, rays_d)


def ndc_rays_camera(H, W, camera_model, near, rays_o, rays_d):
    """NeRF/render.py:376-396: focal lengths from the camera model (gradients reach its intrinsics)."""
    K = camera_model.get_intrinsic()
    return _cf().ndc(H, W, K[0][0], K[1][1], near, rays_o, rays_d)
