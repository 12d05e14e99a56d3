"""Projected-ray-distance loss -- mirror of /root/reference model/ray_dist_loss.py on the fused HIP
kernels (scnerf_prd_loss_fwd / _bwd): same function names, arguments, branches and return values
(`(loss, n_match)` in train mode, `(loss, None)` otherwise), so run_nerf.py:508-598 calls it unchanged.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def preprocess_match(match_result):
    """(model/ray_dist_loss.py:6-19) the matcher's first result -> stacked [2, M, 2] matched key points,
    or (None, None) when there is no match."""
    match_result = match_result[0]
    kps0, kps1, matches = match_result["kps0"], match_result["kps1"], match_result["matches"]
    if len(matches) == 0:
        return None, None
    m = torch.as_tensor(np.asarray(matches), dtype=torch.long, device=kps0.device if torch.is_tensor(kps0) else None)
    kps0, kps1 = torch.as_tensor(kps0), torch.as_tensor(kps1)
    return torch.stack([kps0[m[:, 0]], kps1[m[:, 1]]])


class _PrdLossFunction(torch.autograd.Function):
    """apply(kps0, kps1, rays0_o, rays0_d, rays1_o, rays1_d, K, E2, eps, threshold, negate_fx) -> (loss, n_match)"""

    @staticmethod
    def forward(ctx, kps0, kps1, r0o, r0d, r1o, r1d, K, E2, eps, threshold, negate_fx):
        args = [t.contiguous().float() for t in (kps0, kps1, r0o, r0d, r1o, r1d, K, E2)]
        loss, n_match, sums = ops.prd_loss_fwd(*args, eps, threshold, negate_fx, False)
        ctx.save_for_backward(*args, sums)
        ctx.consts = (eps, threshold, negate_fx)
        ctx.mark_non_differentiable(n_match)
        return loss, n_match

    @staticmethod
    def backward(ctx, g_loss, _g_n):
        *args, sums = ctx.saved_tensors
        eps, threshold, negate_fx = ctx.consts
        g = ops.prd_loss_bwd(*args, eps, threshold, negate_fx, sums, g_loss.float())
        need = ctx.needs_input_grad
        return (None, None) + tuple(g[i] if need[i + 2] else None for i in range(6)) + (None, None, None)


def proj_ray_dist_loss_single(kps0_list, kps1_list, img_idx0, img_idx1, rays0, rays1, mode, device, H, W, args,
                              camera_model=None, intrinsic=None, extrinsic=None, eps=1e-10, i_map=None,
                              method="NeRF", _sync=True):
    """(model/ray_dist_loss.py:22-246)  kps*_list [M,2] matched key points of images img_idx0 / img_idx1,
    rays0 / rays1 = (rays_o, rays_d) [M,3] of those key points.  Parameter sources per mode as the
    reference: train + camera_model -> its current K and the two poses looked up through `i_map`
    (:51-64); train without -> the given (noisy) intrinsic / extrinsic (:66-75); val / test -> the
    ground-truth extrinsic, K from the camera model if there is one (:77-95).

    `_sync` (not in the reference): the reference returns the match count as a python float (:223-227) -- a host read of a device
    scalar in the middle of the step, behind which the GPU idles until the backward pass has been launched (~1 ms of a 12 ms step).
    run_nerf.py only logs the number.  `_sync=False` returns it as a 0-dim device tensor instead, for loops that do not want the
    stall (bench.py --config 3); the default keeps the reference's return type."""
    assert mode in ["train", "val", "test"]
    assert method in ["NeRF", "NeRF++"]
    # (:47-50) `kps*_list[:, 0].max() < W`, `[:, 1].max() < H`: four host reads of device scalars in the reference; here
    # through the deferred check of the ray generator (verdict read at the next call / checkpoint / exit)
    from .get_rays import KEYPOINT_CHECK
    KEYPOINT_CHECK.submit(torch.as_tensor(kps0_list), H, W, lower=False, what="projected-ray-distance")
    KEYPOINT_CHECK.submit(torch.as_tensor(kps1_list), H, W, lower=False, what="projected-ray-distance")
    if mode == "train" and camera_model is not None:
        assert intrinsic is None and extrinsic is None and i_map is not None
        intrinsic = camera_model.get_intrinsic()
        c0 = int(np.where(np.asarray(i_map) == img_idx0)[0][0])
        c1 = int(np.where(np.asarray(i_map) == img_idx1)[0][0])
        # (two views + one stack: indexing with the python list [c0, c1] builds an index tensor on the host and copies it
        #  over -- from pageable memory that copy waits for everything queued on the stream, a host sync in the middle of the step)
        E = camera_model.get_extrinsic()
        extrinsic = torch.stack([E[c0], E[c1]])
    else:
        assert extrinsic is not None
        if camera_model is not None:
            assert intrinsic is None
            intrinsic = camera_model.get_intrinsic()
        assert intrinsic is not None
        if mode == "train":
            assert isinstance(intrinsic, torch.Tensor) and isinstance(extrinsic, torch.Tensor)
        extrinsic = torch.as_tensor(extrinsic)
        extrinsic = torch.stack([extrinsic[img_idx0], extrinsic[img_idx1]])
    dev = rays0[0].device
    K = torch.as_tensor(intrinsic).to(dev)
    E2 = extrinsic.to(dev)
    kps0 = torch.as_tensor(kps0_list).to(dev)
    kps1 = torch.as_tensor(kps1_list).to(dev)
    negate_fx = method == "NeRF"
    threshold = float(args.proj_ray_dist_threshold)
    if mode == "train":
        loss, n_match = _PrdLossFunction.apply(kps0, kps1, rays0[0], rays0[1], rays1[0], rays1[1], K, E2,
                                               float(eps), threshold, negate_fx)
        return loss, (n_match.item() if _sync else n_match)      # the reference returns a python float (:226-229)
    with torch.no_grad():
        tens = [t.contiguous().float() for t in (kps0, kps1, rays0[0], rays0[1], rays1[0], rays1[1], K, E2)]
        loss, _, _ = ops.prd_loss_fwd(*tens, float(eps), threshold, negate_fx, True)
    return loss, None
