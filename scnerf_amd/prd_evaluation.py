"""Mirror of the reference's projected-ray-distance *evaluation* (/root/reference model/prd_evaluation.py):
`projected_ray_distance_evaluation` (:66-183) and `filter_matches_with_gt` (:189-332), same names, arguments
and return values, so `run_nerf.py:693-731, :879-914, :912-951` call it unchanged
(`scnerf_amd.dropin.install()` registers it as `prd_evaluation`).

What runs where: the per-match arithmetic -- closest points of the two rays, re-projection through the
ground-truth cameras, the 1-pixel^2 / chirality filter, the evaluation-mode loss -- is the PRD kernels
(csrc/prd_loss.hip: scnerf_prd_filter, scnerf_prd_loss_fwd); the rays come from the ray-generator kernels
through `ray_fun` / `ray_fun_gt` (the mirrored get_rays_kps_*).  The key-point MATCHERS (SuperGlue / SIFT,
`model/reprojection.py`) and the pair selection are host-side and stay the reference's: they are looked up at
call time (`reprojection` or `model.reprojection`, whichever the caller's sys.path resolves), never imported
by this package."""
from __future__ import annotations

import importlib

import numpy as np
import torch

from . import ops
from .ray_dist_loss import preprocess_match, proj_ray_dist_loss_single

tol = 1e-4
match_num = 4


def _matchers():
    """(runSuperGlueSinglePair, runSIFTSinglePair, image_pair_candidates) of the reference's reprojection module."""
    last = None
    for name in ("reprojection", "model.reprojection"):
        try:
            m = importlib.import_module(name)
            return m.runSuperGlueSinglePair, m.runSIFTSinglePair, m.image_pair_candidates
        except ImportError as e:          # the module, or one of ITS dependencies (cv2, SuperGlue), is missing
            last = e
    raise ImportError("the key-point matchers of the reference (model/reprojection.py) are not importable: %s" % last)


def filter_matches_with_gt(kps0_list, kps1_list, W, H, gt_intrinsic, gt_extrinsic, rays0, rays1, args, method, device,
                           eps=1e-6):
    """bool [M]: matches whose two re-projections through the ground-truth cameras (gt_extrinsic [2,4,4]) land
    within one squared pixel of the detected key points and whose closest points are in front of both cameras."""
    assert method in ["NeRF", "NeRF++"]
    assert kps0_list.dim() == 2 and kps1_list.dim() == 2
    dev = rays0[0].device
    tens = [torch.as_tensor(t).to(dev).contiguous().float()
            for t in (kps0_list, kps1_list, rays0[0], rays0[1], rays1[0], rays1[1], gt_intrinsic, gt_extrinsic)]
    return ops.prd_filter(*tens, float(eps), 1.0, method == "NeRF")


def projected_ray_distance_evaluation(images, index_list, args, ray_fun, ray_fun_gt, H, W, mode, matcher, gt_intrinsic,
                                      gt_extrinsic, method, device, intrinsic=None, extrinsic=None, camera_model=None,
                                      i_map=None):
    """Mean projected ray distance over the feasible image pairs of `index_list`.  mode "train": the cameras
    under training (camera model parameters, or the given noisy intrinsic / extrinsic); "val" / "test": matches
    are first filtered with the ground truth, rays use the ground-truth poses."""
    run_superglue, run_sift, image_pair_candidates = _matchers()
    match_fun = run_superglue if args.matcher == "superglue" else run_sift
    with torch.no_grad():
        pairs = image_pair_candidates(gt_extrinsic[index_list].cpu().numpy(), args, index_list)
    distances = []
    for img_i in pairs.keys():
        for img_j in pairs[img_i]:
            if img_i >= img_j:
                continue
            kps0_list, kps1_list = preprocess_match(match_fun(matcher, images[img_i], images[img_j], 0, args))
            if kps0_list is None and kps1_list is None:
                continue
            if mode != "train":
                rays_i_gt = ray_fun_gt(H=H, W=W, focal=gt_intrinsic[0][0], extrinsic=gt_extrinsic[img_i], kps_list=kps0_list)
                rays_j_gt = ray_fun_gt(H=H, W=W, focal=gt_intrinsic[0][0], extrinsic=gt_extrinsic[img_j], kps_list=kps1_list)
                keep = filter_matches_with_gt(kps0_list=kps0_list, kps1_list=kps1_list, H=H, W=W, gt_intrinsic=gt_intrinsic,
                                              gt_extrinsic=gt_extrinsic[[img_i, img_j]], rays0=rays_i_gt, rays1=rays_j_gt,
                                              args=args, device=device, method=method)
                keep = keep.to(kps0_list.device)
                kps0_list, kps1_list = kps0_list[keep], kps1_list[keep]
            if camera_model is None:
                poses = gt_extrinsic if mode != "train" else extrinsic
                rays_i = ray_fun(H=H, W=W, focal=intrinsic[0][0], extrinsic=poses[img_i], kps_list=kps0_list)
                rays_j = ray_fun(H=H, W=W, focal=intrinsic[0][0], extrinsic=poses[img_j], kps_list=kps1_list)
                dist, _ = proj_ray_dist_loss_single(kps0_list=kps0_list, kps1_list=kps1_list, img_idx0=img_i,
                                                    img_idx1=img_j, rays0=rays_i, rays1=rays_j, mode=mode, device=device,
                                                    H=H, W=W, args=args, intrinsic=gt_intrinsic, extrinsic=poses)
            else:
                held_out = mode != "train"
                slot_i = None if held_out else np.where(np.asarray(i_map) == img_i)[0][0]
                slot_j = None if held_out else np.where(np.asarray(i_map) == img_j)[0][0]
                rays_i = ray_fun(H=H, W=W, camera_model=camera_model, extrinsic=gt_extrinsic[img_i] if held_out else None,
                                 kps_list=kps0_list, idx_in_camera_param=slot_i)
                rays_j = ray_fun(H=H, W=W, camera_model=camera_model, extrinsic=gt_extrinsic[img_j] if held_out else None,
                                 kps_list=kps1_list, idx_in_camera_param=slot_j)
                dist, _ = proj_ray_dist_loss_single(kps0_list=kps0_list, kps1_list=kps1_list, img_idx0=img_i,
                                                    img_idx1=img_j, rays0=rays_i, rays1=rays_j, mode=mode, device=device,
                                                    H=H, W=W, args=args, i_map=i_map, camera_model=camera_model,
                                                    extrinsic=gt_extrinsic if held_out else None)
            if not torch.isnan(dist):
                distances.append(dist.item())
    return torch.tensor(distances).mean()
