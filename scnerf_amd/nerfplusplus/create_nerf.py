"""nerfplusplus/create_nerf.py: cascade of NerfNetWithAutoExpo models, the camera model, the optimizer and
checkpoint reload, with the reference's structure and return values `(start, models, camera_model)`.

Differences a maintainer should know: the networks are wrapped in a parameter-less `_Replica` container
instead of gloo `DistributedDataParallel` (so state-dict keys keep the `module.` prefix): gradients are
synchronised by ONE RCCL all-reduce per step over flat buffers (scnerf_amd.parallel.FlatGradAllReduce,
`models['grad_sync']`), which -- unlike the reference (:64-65) -- also covers the camera parameters."""
from __future__ import annotations

import json
import logging
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from ..camera_model import (PinholeModelRotNoiseLearning10kRayoRayd,
                            PinholeModelRotNoiseLearning10kRayoRaydDistortion)
from ..optim import CustomAdamOptimizer, FusedAdam
from .ddp_model import NerfNetWithAutoExpo

logger = logging.getLogger(__package__)


class _Replica(nn.Module):
    """Stands where DistributedDataParallel stands in the reference: `.module`, `module.`-prefixed keys."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


def create_nerf(rank, args, camera_info):
    """(create_nerf.py:14-154)"""
    torch.manual_seed(777)
    device = torch.device("cuda", rank) if isinstance(rank, int) else torch.device(rank)
    torch.cuda.set_device(device)
    camera_model = None
    H = W = None
    if args.use_camera:
        intrinsics, extrinsics = camera_info["intrinsics"], camera_info["extrinsics"]
        H, W = camera_info["H"], camera_info["W"]
        if args.camera_model == "pinhole_rot_noise_10k_rayo_rayd":
            camera_model = PinholeModelRotNoiseLearning10kRayoRayd(intrinsics, extrinsics, args, H, W).to(device)
        else:
            camera_model = PinholeModelRotNoiseLearning10kRayoRaydDistortion(
                intrinsics, extrinsics, args, H, W, camera_info["k"]).to(device)

    models = OrderedDict()
    models['cascade_level'] = args.cascade_level
    models['cascade_samples'] = [int(x.strip()) for x in args.cascade_samples.split(',')]
    parameters = []
    for m in range(models['cascade_level']):
        img_names = None
        if args.optim_autoexpo:
            with open(os.path.join(args.basedir, args.expname, 'train_images.json')) as file:
                img_names = json.load(file)
        net = NerfNetWithAutoExpo(args, optim_autoexpo=args.optim_autoexpo, img_names=img_names).to(device)
        # contiguous parameter buffers before the optimizer looks at the tensors (one fused-Adam segment each)
        net.nerf_net.fg_net.flat_parameters()
        net.nerf_net.bg_net.flat_parameters()
        net = _Replica(net)
        parameters = [*parameters, *net.parameters()]
        models['net_{}'.format(m)] = net
    if camera_model is not None:
        parameters = [*parameters, *camera_model.parameters()]

    if args.use_custom_optim:
        optim = CustomAdamOptimizer(params=parameters, lr=args.lrate, betas=(0.9, 0.999),
                                    weight_decay=args.non_linear_weight_decay, H=H, W=W, args=args)
    else:
        optim = FusedAdam(parameters, lr=args.lrate)
    models["optim"] = optim

    start = -1
    if (args.ckpt_path is not None) and (os.path.isfile(args.ckpt_path)):
        ckpts = [args.ckpt_path]
    else:
        d = os.path.join(args.basedir, args.expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith('.pth')] if os.path.isdir(d) else []

    def path2iter(path):
        tmp = os.path.basename(path)[:-4]
        return int(tmp[tmp.rfind('_') + 1:])

    ckpts = sorted(ckpts, key=path2iter)
    logger.info('Found ckpts: {}'.format(ckpts))
    if len(ckpts) > 0 and not args.no_reload:
        fpath = ckpts[-1]
        logger.info('Reloading from: {}'.format(fpath))
        start = path2iter(fpath)
        to_load = torch.load(fpath, map_location=device)
        for m in range(models['cascade_level']):
            name = 'net_{}'.format(m)
            models[name].load_state_dict(to_load[name])
        model_dict = models["optim"].state_dict()
        model_dict["state"].update(to_load["optim"]["state"])
        models["optim"].load_state_dict(model_dict)
        if getattr(args, "load_camera", False):
            assert not args.load_test
            camera_origin = camera_model.state_dict()
            camera_origin.update({k: v for k, v in to_load["camera_model"].items()
                                  if k not in ["extrinsics_noise", "extrinsics_initial"]})
            camera_model.load_state_dict(camera_origin)
        if getattr(args, "load_test", False):
            assert not args.load_camera
            camera_origin = camera_model.state_dict()
            camera_origin.update(to_load["camera_model"])
            camera_model.load_state_dict(camera_origin)

    if not getattr(args, "load_test", False) and args.use_camera:
        if start < args.add_ie and hasattr(camera_model, "intrinsics_noise") and hasattr(camera_model, "extrinsics_noise"):
            camera_model.intrinsics_noise.requires_grad_(False)
            camera_model.extrinsics_noise.requires_grad_(False)
            logger.info("Deactivated learnable intrinsic and extrinsic")
        if start < args.add_radial and hasattr(camera_model, "distortion_noise"):
            camera_model.distortion_noise.requires_grad_(False)
            logger.info("Deactivated learnable radial distortion")
        if start < args.add_od and hasattr(camera_model, "ray_o_noise") and hasattr(camera_model, "ray_d_noise"):
            camera_model.ray_o_noise.requires_grad_(False)
            camera_model.ray_d_noise.requires_grad_(False)
            logger.info("Deactivated learnable ray offset and direction noise")
    return start, models, camera_model
