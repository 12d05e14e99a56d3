"""Mirror of the reference's `create_nerf` module (/root/reference NeRF/create_nerf.py) for the
render path: `run_network`, `batchify`, `create_nerf` (the render_kwargs contract, :71-93).

The reference hides the embedders inside an opaque lambda (`network_query_fn`, :67-69); here it
is a `FusedNetworkQuery` object that carries the embedder configuration so `render_rays` can check
that the fused kernels (multires 10 / 4 compiled in) match what the caller asked for, and that can
still be *called* like the reference's closure: query(pts, viewdirs, network_fn) -> raw."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import mlp_layout as ML
from . import ops
from .run_nerf_helpers import NeRF, get_embedder


def _unwrap(net):
    return net.module if isinstance(net, nn.DataParallel) else net


class _QueryFunction(torch.autograd.Function):
    """raw = network(pts, viewdirs) as one differentiable node (fused PE + MLP)."""

    @staticmethod
    def forward(ctx, pts, viewdirs, net, track, *params):
        shape = pts.shape
        spr = int(shape[-2]) if pts.dim() >= 3 else 1
        p = pts.reshape(-1, 3).contiguous().float()
        v = viewdirs.reshape(-1, 3).contiguous().float()
        if p.shape[0] != v.shape[0] * spr:
            raise ValueError("pts [..., S, 3] and viewdirs [..., 3] disagree: %s vs %s" % (tuple(shape), tuple(viewdirs.shape)))
        # (`track`: torch.is_grad_enabled() at the call; needs_input_grad alone stays True under torch.no_grad())
        train = track and any(ctx.needs_input_grad)
        flat = net.flat_parameters()
        save = ops.save_workspace(p.shape[0], p.device) if train else None
        # the arithmetic in force (ops.mlp_arithmetic), as in render_rays: the resident kernels' streams + the chunk maxima
        # their weight-gradient GEMMs scale by, or the fused fp32-MFMA kernels
        if train or p.shape[0] == 0:
            wf = ops.pack_weights(flat, "fwd")
            planes = ops.pack_for_arithmetic(flat, train) if p.shape[0] > 0 else None
        else:
            wf, planes = ops.inference_packs(net, flat)       # (forward-only: packed once per weight version)
        maxima = ops.ChunkMaxima(p.shape[0], p.device) if (train and isinstance(planes, ops.ResidentWeights)) else None
        raw = ops.mlp_fwd(p, v, spr, wf, save, planes=planes, maxima=maxima)
        ctx.state = (p, v, spr, save, ops.pack_weights(flat, "bwd") if train else None, shape, viewdirs.shape, planes, maxima)
        return raw.view(*shape[:-1], 4)

    @staticmethod
    def backward(ctx, g_raw):
        p, v, spr, save, wbk, shape, vshape, planes, maxima = ctx.state
        d_raw = g_raw.reshape(-1, 4).contiguous().float()
        grads, d_pts, d_views = ops.mlp_bwd(d_raw, p, v, spr, wbk, save, planes=planes, maxima=maxima)
        flat_grad = ops.nerf_wgrad(save, grads, d_raw, p.shape[0], maxima=maxima)
        d_v = d_views.view(-1, spr, 3).sum(1).view(vshape)
        gs = [flat_grad[ML.PARAM_OFFSETS[n]: ML.PARAM_OFFSETS[n] + int(torch.Size(s).numel())].view(s)
              for n, s in ML.PARAM_SHAPES]
        ctx.state = None
        return (d_pts.view(shape), d_v, None, None, *gs)


def _fused_applies(net, embed_fn, embeddirs_fn, viewdirs_given=True) -> bool:
    """the one configuration the fused kernels are built for: the standard network with multires 10 / 4 encodings"""
    return (isinstance(net, NeRF) and net.is_standard() and viewdirs_given
            and getattr(embed_fn, "num_freqs", None) == ML.L_PTS and getattr(embeddirs_fn, "num_freqs", None) == ML.L_VIEWS)


def run_network(inputs, viewdirs, fn, embed_fn=None, embeddirs_fn=None, netchunk=1024 * 64):
    """Prepares inputs and applies network `fn` (reference :18-32).  inputs [N, S, 3], viewdirs
    [N, 3] -> [N, S, 4].  Standard network + encodings: the encodings and the MLP run as one fused kernel, and
    `netchunk` (a memory workaround of the reference) is accepted and ignored.  Anything else -- another network
    shape, other encoding widths, no view directions, a foreign module -- goes exactly the reference's way: embed
    (the stand-alone encoding kernel), concatenate, `batchify(fn, netchunk)`."""
    net = _unwrap(fn)
    if _fused_applies(net, embed_fn, embeddirs_fn, viewdirs is not None):
        return _QueryFunction.apply(inputs, viewdirs, net, torch.is_grad_enabled(), *net.ordered_parameters())
    inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
    embedded = embed_fn(inputs_flat)
    if viewdirs is not None:
        input_dirs = viewdirs[:, None].expand(inputs.shape)
        embedded = torch.cat([embedded, embeddirs_fn(torch.reshape(input_dirs, [-1, input_dirs.shape[-1]]).contiguous())], -1)
    outputs_flat = batchify(fn, netchunk)(embedded)
    return torch.reshape(outputs_flat, list(inputs.shape[:-1]) + [outputs_flat.shape[-1]])


class FusedNetworkQuery:
    """Callable stand-in for the reference's `network_query_fn` closure (:67-69): carries the embedders so that
    render_rays can tell whether the fused kernels apply (`fused_for(net)`) and can otherwise call it like any
    other closure."""
    is_fused_query = True

    def __init__(self, embed_fn, embeddirs_fn, netchunk=None):
        self.embed_fn, self.embeddirs_fn, self.netchunk = embed_fn, embeddirs_fn, netchunk

    def fused_for(self, net) -> bool:
        return _fused_applies(_unwrap(net), self.embed_fn, self.embeddirs_fn)

    def check(self, net: NeRF):
        """raises unless the fused kernels apply to `net` with these encodings"""
        net.require_standard()
        if not self.fused_for(net):
            raise NotImplementedError("fused kernels are built for multires %d / multires_views %d"
                                      % (ML.L_PTS, ML.L_VIEWS))

    def __call__(self, inputs, viewdirs, network_fn):
        return run_network(inputs, viewdirs, network_fn, self.embed_fn, self.embeddirs_fn,
                           self.netchunk if self.netchunk is not None else 1024 * 64)


def batchify(fn, chunk):
    """Kept for API parity (reference :187-196); the fused kernel needs no chunking."""
    if chunk is None or chunk <= 0:           # (0: `netchunk_per_gpu` absent from args -- no chunking, not range(0, N, 0))
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret


def create_nerf(args, noisy_focal, noisy_poses, H, W, mode="train", device="cuda"):
    """Instantiate the coarse / fine networks, the query object, the render kwargs, the camera model
    and the optimizer with the reference's structure and return order (:34-184):
    (render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer, camera_model).
    Checkpoints are found and restored as the reference does (:142-172): `args.ft_path`, else the last
    `*tar*` file in `basedir/expname`, unless `args.no_reload`; the optimizer's per-parameter state is
    merged into the fresh optimizer's, the networks load their `module.`-prefixed keys, the camera
    model its own state."""
    camera_model = None
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    input_ch_views = 0
    embeddirs_fn = None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    model = NeRF(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=skips,
                 input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs)
    model = nn.DataParallel(model).to(device)       # container only: keeps the 'module.' checkpoint keys
    grad_vars = list(model.parameters())
    model_fine = None
    if args.N_importance > 0:
        model_fine = NeRF(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch,
                          skips=skips, input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs)
        model_fine = nn.DataParallel(model_fine).to(device)
        grad_vars += list(model_fine.parameters())

    network_query_fn = FusedNetworkQuery(embed_fn, embeddirs_fn,
                                         getattr(args, "netchunk_per_gpu", 0) * getattr(args, "n_gpus", 1))
    render_kwargs_train = {
        'network_query_fn': network_query_fn, 'perturb': args.perturb, 'N_importance': args.N_importance,
        'network_fine': model_fine, 'N_samples': args.N_samples, 'network_fn': model,
        'use_viewdirs': args.use_viewdirs, 'white_bkgd': args.white_bkgd, 'raw_noise_std': args.raw_noise_std,
    }
    if args.dataset_type != 'llff' or args.no_ndc:
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.

    if args.camera_model != "none":
        from .camera_dict import camera_dict
        cw_init, ch_init = W / 2, H / 2
        fx_init = W if args.run_without_colmap != "none" else noisy_focal
        fy_init = H if args.run_without_colmap != "none" else noisy_focal
        intrinsic_init = torch.tensor([[fx_init, 0, cw_init, 0], [0, fy_init, ch_init, 0],
                                       [0, 0, 1, 0], [0, 0, 0, 1]])
        with torch.no_grad():
            camera_model = camera_dict[args.camera_model](
                intrinsics=intrinsic_init, extrinsics=noisy_poses, args=args, H=H, W=W).to(device)
        grad_vars += list(camera_model.parameters())

    # one contiguous buffer per network *before* the optimizer looks at the tensors, so that each
    # network becomes a single fused-Adam segment
    _unwrap(model).flat_parameters()
    if model_fine is not None:
        _unwrap(model_fine).flat_parameters()
    from .optim import CustomAdamOptimizer, FusedAdam
    if getattr(args, "use_custom_optim", False):
        optimizer = CustomAdamOptimizer(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999),
                                        weight_decay=args.non_linear_weight_decay, H=H, W=W, args=args)
    else:
        optimizer = FusedAdam(grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    # one process per GPU (torch.distributed initialised by the launcher): the optimizer's step() first
    # all-reduces its gradient arena -- networks AND camera parameters, one collective (parallel.py)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        from .parallel import FlatGradAllReduce
        optimizer.grad_sync = FlatGradAllReduce.for_optimizer(optimizer, dist.get_world_size())
    start = 0
    ft_path = getattr(args, "ft_path", None)
    basedir, expname = getattr(args, "basedir", None), getattr(args, "expname", None)
    if ft_path is not None and ft_path != 'None':
        ckpts = [ft_path]
    elif basedir is not None and expname is not None and os.path.isdir(os.path.join(basedir, expname)):
        ckpts = [os.path.join(basedir, expname, f) for f in sorted(os.listdir(os.path.join(basedir, expname)))
                 if 'tar' in f]
    else:
        ckpts = []
    print('Found ckpts', ckpts)
    if len(ckpts) > 0 and not getattr(args, "no_reload", False):
        ckpt_path = ckpts[-1]
        print('Reloading from', ckpt_path)
        ckpt = torch.load(ckpt_path, map_location=device)
        start = ckpt['global_step']
        optim_dict = optimizer.state_dict()
        optim_dict["state"].update(ckpt['optimizer_state_dict']["state"])
        optimizer.load_state_dict(optim_dict)
        model.load_state_dict(ckpt['network_fn_state_dict'])
        if model_fine is not None:
            model_fine.load_state_dict(ckpt['network_fine_state_dict'])
        if camera_model is not None and "camera_model" in ckpt.keys():
            camera_model.load_state_dict(ckpt["camera_model"])
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer, camera_model
