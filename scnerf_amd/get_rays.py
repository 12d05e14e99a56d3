"""Mirror of the reference's `get_rays` module (/root/reference NeRF/get_rays.py) plus the two NDC
warps of NeRF/render.py:357-396; the arithmetic runs in the HIP ray-generator kernels
(scnerf_amd/csrc/camera_rays.hip) through scnerf_amd.camera_functional."""
from __future__ import annotations

import numpy as np
import torch


def _cf():
    from . import camera_functional
    return camera_functional


def get_rays_full_image_no_camera(H, W, focal, extrinsic):
    """Pinhole rays of every pixel of one image (reference :5-23)."""
    assert extrinsic.dim() == 2
    return _cf().pinhole_rays(H, W, focal, extrinsic, None)


class _DeferredKeypointCheck:
    """The reference asserts `kps_list[:, 0].max() < W` and `[:, 1].max() < H` on GPU tensors (get_rays.py:78-79,
    :109-110): two host reads of device scalars, each of which drains the queue at the very start of a step --
    the GPU then idles while the host re-issues the ~25 small launches in front of the first network kernel
    (~0.5 ms of a 30 ms step).  Here the verdict (upper AND lower bound, one reduction) is copied to pinned host
    memory asynchronously and looked at when the NEXT ray-generation call comes in -- by then it has long
    arrived, so nothing waits.  An out-of-range batch therefore raises the reference's AssertionError one call
    late; the kernels clamp the pixel they read the noise grids at, so the late report is the only
    difference.  `flush()` waits for whatever is pending: the host layer calls it where a training loop reaches a
    boundary anyway -- optimizer checkpoints (FusedAdam.state_dict), full-image renders (render_path) -- and at
    interpreter exit (reported on stderr there), so the last batches of a run are checked too; the message names the
    call it belongs to.  CPU tensors are checked at once.  SCNERF_SYNC_KEYPOINT_CHECK=1 (or `.synchronous = True`)
    restores the reference's behaviour -- the assertion is raised by the faulty call itself, at the price of the host
    reads -- for scripts that catch it around the call."""

    def __init__(self):
        import os
        self.pending = []          # (event, pinned flag, message)
        self.carried = None        # a failure found at a checkpoint boundary, raised by the NEXT poll (see carry())
        self.calls = 0
        self.synchronous = os.environ.get("SCNERF_SYNC_KEYPOINT_CHECK", "0") not in ("", "0")

    def _raise_if_set(self, flag, message):
        assert not bool(flag.item()), message

    def carry(self, message):
        """A failure the caller could not raise where it found it (FusedAdam.state_dict: the checkpoint is written first):
        it stays pending and the next poll -- the next ray-generation call, render_path, interpreter exit -- raises it."""
        if self.carried is None:
            self.carried = message

    def poll(self, block=False):
        waiting, self.pending = self.pending, []
        failed, self.carried = self.carried, None
        for ev, flag, message in waiting:
            if block:
                ev.synchronize()
            if not ev.query():
                self.pending.append((ev, flag, message))
            elif bool(flag.item()) and failed is None:
                failed = message                    # reported once; the remaining entries are still processed
        assert failed is None, failed

    def flush(self):
        self.poll(block=True)

    def flush_at_exit(self):
        import sys
        try:
            self.flush()
        except AssertionError as e:             # (raising inside atexit would only print a traceback after the fact)
            sys.stderr.write("scnerf_amd.get_rays: %s (found at interpreter exit)\n" % e)
        except Exception:
            pass

    def submit(self, kps_list, H, W, lower=True, what="ray-generation"):
        """`lower=False`: the upper bounds only (what the reference's projected-ray-distance term asserts,
        model/ray_dist_loss.py:47-50)."""
        self.poll()
        self.calls += 1
        if kps_list.numel() == 0:
            return
        xy = kps_list[:, :2]
        limit = torch.tensor([W, H], dtype=xy.dtype, device="cpu") if not xy.is_cuda else \
            self._limit(W, H, xy)
        bad = ((xy >= limit) | (xy < 0)).any() if lower else (xy >= limit).any()
        message = "key points outside the %d x %d image (%s call #%d of this process)" % (W, H, what, self.calls)
        if not bad.is_cuda or self.synchronous:
            self._raise_if_set(bad, message)
            return
        host = torch.empty((), dtype=torch.bool, device="cpu", pin_memory=True)      # (explicit: the reference script makes CUDA the default tensor type)
        host.copy_(bad, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((ev, host, message))
        if len(self.pending) > 64:                  # never grows without bound
            self.flush()

    _limits = {}

    def _limit(self, W, H, like):
        key = (W, H, like.dtype, str(like.device))
        if key not in self._limits:
            self._limits[key] = torch.tensor([W, H], dtype=like.dtype, device=like.device)
        return self._limits[key]


KEYPOINT_CHECK = _DeferredKeypointCheck()
import atexit as _atexit  # noqa: E402
_atexit.register(KEYPOINT_CHECK.flush_at_exit)


def _assert_inside_image(kps_list, H, W):
    KEYPOINT_CHECK.submit(kps_list, H, W)


def get_rays_kps_no_camera(H, W, focal, extrinsic, kps_list):
    """Pinhole rays at integer pixel coordinates kps_list [N, >=2] (x, y, ...) (reference :75-90)."""
    _assert_inside_image(kps_list, H, W)
    assert extrinsic.dim() == 2
    return _cf().pinhole_rays(H, W, focal, extrinsic, kps_list)


def get_rays_full_image_use_camera(H, W, camera_model, idx_in_camera_param=None, extrinsic=None):
    """Rays of every pixel through the learnable camera model (reference :26-72)."""
    return _cf().camera_rays(H, W, camera_model, None, idx_in_camera_param, extrinsic)


def get_rays_kps_use_camera(H, W, camera_model, kps_list, idx_in_camera_param=None, extrinsic=None):
    """Rays at key points kps_list [N,2] (x, y) through the learnable camera model (reference :93-148)."""
    _assert_inside_image(kps_list, H, W)
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
