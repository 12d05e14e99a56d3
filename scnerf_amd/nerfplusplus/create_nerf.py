"""nerfplusplus/create_nerf.py: cascade of NerfNetWithAutoExpo models, the camera model, the optimizer and
checkpoint reload, with the reference's structure and return values `(start, models, camera_model)`.

Differences a maintainer should know: the networks are wrapped in a parameter-less `_Replica` container
instead of gloo `DistributedDataParallel` (so state-dict keys keep the `module.` prefix).  Gradients are
synchronised by ONE all-reduce per step of the optimizer's flat gradient arena
(scnerf_amd.parallel.FlatGradAllReduce.for_optimizer), which -- unlike the reference (:64-65) -- also covers the
camera parameters: when torch.distributed is initialised with more than one rank, `create_nerf` attaches the
reducer as `models['optim'].grad_sync` (and `models['grad_sync']`), and `optim.step()` runs the collective before
the update -- the training loop's `zero_grad() / backward() / step()` (ddp_train_nerf.py:400-470) stays as it is.
Every rank must take the same curriculum steps (`requires_grad_` toggles change the arena's layout)."""
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


def _world_size():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _build_camera(args, camera_info, device):
    """(create_nerf.py:19-33) the learnable camera model, or None when `use_camera` is off."""
    if not args.use_camera:
        return None, None, None
    H, W = camera_info["H"], camera_info["W"]
    common = (camera_info["intrinsics"], camera_info["extrinsics"], args, H, W)
    if args.camera_model == "pinhole_rot_noise_10k_rayo_rayd":
        cam = PinholeModelRotNoiseLearning10kRayoRayd(*common)
    else:
        cam = PinholeModelRotNoiseLearning10kRayoRaydDistortion(*common, camera_info["k"])
    return cam.to(device), H, W


def _checkpoint_step(path):
    stem = os.path.basename(path)[:-4]            # model_000123.pth -> 123
    return int(stem[stem.rfind('_') + 1:])


def _find_checkpoints(args):
    """(create_nerf.py:80-97) an explicit `ckpt_path`, else every `*.pth` of the experiment directory,
    ordered by the step number in the file name."""
    if args.ckpt_path is not None and os.path.isfile(args.ckpt_path):
        found = [args.ckpt_path]
    else:
        folder = os.path.join(args.basedir, args.expname)
        found = [os.path.join(folder, f) for f in sorted(os.listdir(folder)) if f.endswith('.pth')] \
            if os.path.isdir(folder) else []
    return sorted(found, key=_checkpoint_step)


def _restore(path, models, camera_model, args, device):
    """(create_nerf.py:98-129) networks, the optimizer's per-parameter state (merged into the fresh one), and
    the camera -- everything but the extrinsics with `load_camera`, everything with `load_test`."""
    blob = torch.load(path, map_location=device)
    for m in range(models['cascade_level']):
        key = 'net_{}'.format(m)
        models[key].load_state_dict(blob[key])
    merged = models["optim"].state_dict()
    merged["state"].update(blob["optim"]["state"])
    models["optim"].load_state_dict(merged)
    load_camera, load_test = getattr(args, "load_camera", False), getattr(args, "load_test", False)
    assert not (load_camera and load_test)
    if load_camera or load_test:
        skip = ("extrinsics_noise", "extrinsics_initial") if load_camera else ()
        state = camera_model.state_dict()
        state.update({k: v for k, v in blob["camera_model"].items() if k not in skip})
        camera_model.load_state_dict(state)


def _freeze_for_curriculum(camera_model, start, args):
    """(create_nerf.py:131-154) parameter groups whose activation step (add_ie / add_radial / add_od) lies
    ahead of `start` begin frozen; the training loop switches them on."""
    groups = ((args.add_ie, ("intrinsics_noise", "extrinsics_noise"), "learnable intrinsic and extrinsic"),
              (args.add_radial, ("distortion_noise",), "learnable radial distortion"),
              (args.add_od, ("ray_o_noise", "ray_d_noise"), "learnable ray offset and direction noise"))
    for first_step, names, what in groups:
        if start < first_step and all(hasattr(camera_model, n) for n in names):
            for n in names:
                getattr(camera_model, n).requires_grad_(False)
            logger.info("Deactivated " + what)


def create_nerf(rank, args, camera_info):
    """(create_nerf.py:14-154) -> (start, models, camera_model); `models` = OrderedDict(cascade_level,
    cascade_samples, net_0 .. net_{L-1}, optim)."""
    torch.manual_seed(777)
    device = torch.device("cuda", rank) if isinstance(rank, int) else torch.device(rank)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    camera_model, H, W = _build_camera(args, camera_info, device)

    models = OrderedDict()
    models['cascade_level'] = args.cascade_level
    models['cascade_samples'] = [int(x.strip()) for x in args.cascade_samples.split(',')]
    img_names = None
    if args.optim_autoexpo:
        with open(os.path.join(args.basedir, args.expname, 'train_images.json')) as file:
            img_names = json.load(file)
    parameters = []
    for m in range(models['cascade_level']):
        net = NerfNetWithAutoExpo(args, optim_autoexpo=args.optim_autoexpo, img_names=img_names).to(device)
        # contiguous parameter buffers before the optimizer looks at the tensors (one fused-Adam segment each)
        net.nerf_net.fg_net.flat_parameters()
        net.nerf_net.bg_net.flat_parameters()
        models['net_{}'.format(m)] = _Replica(net)
        parameters += list(models['net_{}'.format(m)].parameters())
    if camera_model is not None:
        parameters += list(camera_model.parameters())
    if args.use_custom_optim:
        models["optim"] = CustomAdamOptimizer(params=parameters, lr=args.lrate, betas=(0.9, 0.999),
                                              weight_decay=args.non_linear_weight_decay, H=H, W=W, args=args)
    else:
        models["optim"] = FusedAdam(parameters, lr=args.lrate)

    world = _world_size()
    if world > 1:
        from ..parallel import FlatGradAllReduce
        models["optim"].grad_sync = FlatGradAllReduce.for_optimizer(models["optim"], world)
        models["grad_sync"] = models["optim"].grad_sync

    start = -1
    ckpts = _find_checkpoints(args)
    logger.info('Found ckpts: {}'.format(ckpts))
    if ckpts and not args.no_reload:
        logger.info('Reloading from: {}'.format(ckpts[-1]))
        start = _checkpoint_step(ckpts[-1])
        _restore(ckpts[-1], models, camera_model, args, device)
    if args.use_camera and not getattr(args, "load_test", False):
        _freeze_for_curriculum(camera_model, start, args)
    return start, models, camera_model
