"""nerfplusplus/ddp_model.py on the fused kernels: `NerfNet.forward` is ONE autograd node -- sample
placement (foreground points + inverted-sphere background points), the two networks (scnerf_mlp_fwd with
pt_dims 3 and 4), the two-level compositing -- with gradients to the rays, the depths and every parameter."""
from __future__ import annotations

import logging
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import _capi
from .. import mlp_layout as ML
from .. import ops
from ..ops import _p, _stream
from .nerf_network import Embedder, MLPNet

logger = logging.getLogger(__package__)
TINY_NUMBER = 1e-6
HUGE_NUMBER = 1e10

_OUT_KEYS = ("rgb", "fg_weights", "bg_weights", "fg_rgb", "fg_depth", "bg_rgb", "bg_depth", "bg_lambda")


def depth2pts_outside(ray_o, ray_d, depth):
    """(ddp_model.py:16-45) ray_o, ray_d [..., 3], depth [...] (inverse distance to the sphere centre) ->
    (pts [..., 4] = unit direction + depth, depth_real [...]).  Forward only (inside NerfNet.forward the
    same kernel runs with its backward)."""
    shape = depth.shape
    o = ray_o.expand(*shape, 3).reshape(-1, 3).contiguous().float()
    d = ray_d.expand(*shape, 3).reshape(-1, 3).contiguous().float()
    z = depth.reshape(-1, 1).contiguous().float()
    n = o.shape[0]
    pts = torch.empty((n, 4), dtype=torch.float32, device=o.device)
    real = torch.empty((n, 1), dtype=torch.float32, device=o.device)
    with torch.no_grad():
        st = _capi.load().scnerf_npp_points_fwd(_p(o), _p(d), None, _p(z), None, _p(pts), None, _p(real), n, 0, 1,
                                                _stream())
    _capi.check(st, "scnerf_npp_points_fwd")
    return pts.view(*shape, 4), real.view(shape)


class _NerfNetFunction(torch.autograd.Function):
    """apply(ray_o, ray_d, fg_z_max, fg_z, bg_z, fg_net, bg_net, *fg_params, *bg_params) -> the 8 outputs
    of NerfNet.forward (_OUT_KEYS order)."""

    @staticmethod
    def forward(ctx, ray_o, ray_d, fg_z_max, fg_z, bg_z, fg_net, bg_net, track, *params):
        sf, sb = fg_z.shape[-1], bg_z.shape[-1]
        dots = ray_d.shape[:-1]
        o = ray_o.reshape(-1, 3).contiguous().float()
        d = ray_d.reshape(-1, 3).contiguous().float()
        zmax = fg_z_max.reshape(-1).contiguous().float()
        zf = fg_z.reshape(-1, sf).contiguous().float()
        zb = bg_z.reshape(-1, sb).contiguous().float()
        n, dev = o.shape[0], o.device
        lib = _capi.load()
        # (`track`: torch.is_grad_enabled() at the call; needs_input_grad alone stays True under torch.no_grad())
        train = track and any(ctx.needs_input_grad)
        fg_pts = torch.empty((n * sf, 3), dtype=torch.float32, device=dev)
        bg_pts = torch.empty((n * sb, 4), dtype=torch.float32, device=dev)
        views = torch.empty((n, 3), dtype=torch.float32, device=dev)
        _capi.check(lib.scnerf_npp_points_fwd(_p(o), _p(d), _p(zf), _p(zb), _p(fg_pts), _p(bg_pts), _p(views), None,
                                              n, sf, sb, _stream()), "scnerf_npp_points_fwd")
        flat_f, flat_b = fg_net.flat_parameters(), bg_net.flat_parameters()
        save_f = ops.save_workspace(n * sf, dev, 3) if train else None
        save_b = ops.save_workspace(n * sb, dev, 4) if train else None
        # the arithmetic in force (ops.mlp_arithmetic): the resident kernels' streams, as in the SCNeRF step
        if train or n == 0:
            pl_f = ops.pack_for_arithmetic(flat_f, train, 3, remap=fg_net.pack_remap()) if n > 0 else None
            pl_b = ops.pack_for_arithmetic(flat_b, train, 4, remap=bg_net.pack_remap()) if n > 0 else None
            wf_f = ops.pack_weights(flat_f, "fwd", pd=3, remap=fg_net.pack_remap())
            wf_b = ops.pack_weights(flat_b, "fwd", pd=4, remap=bg_net.pack_remap())
        else:                                          # (forward-only: packed once per weight version)
            wf_f, pl_f = ops.inference_packs(fg_net, flat_f, 3, remap=fg_net.pack_remap())
            wf_b, pl_b = ops.inference_packs(bg_net, flat_b, 4, remap=bg_net.pack_remap())
        resident = train and isinstance(pl_f, ops.ResidentWeights)
        mx_f = ops.ChunkMaxima(n * sf, dev) if resident else None        # (for the fp16 weight-gradient GEMMs)
        mx_b = ops.ChunkMaxima(n * sb, dev) if resident else None
        raw_f = ops.mlp_fwd(fg_pts, views, sf, wf_f, save_f, pd=3, planes=pl_f, maxima=mx_f)
        raw_b = ops.mlp_fwd(bg_pts, views, sb, wf_b, save_b, pd=4, planes=pl_b, maxima=mx_b)
        out = {"rgb": (n, 3), "fg_weights": (n, sf), "bg_weights": (n, sb), "fg_rgb": (n, 3), "fg_depth": (n,),
               "bg_rgb": (n, 3), "bg_depth": (n,), "bg_lambda": (n,)}
        t = {k: torch.empty(sh, dtype=torch.float32, device=dev) for k, sh in out.items()}
        _capi.check(lib.scnerf_npp_composite_fwd(_p(raw_f), _p(raw_b), _p(zf), _p(zmax), _p(zb), _p(d),
                                                 *[_p(t[k]) for k in _OUT_KEYS], n, sf, sb, _stream()),
                    "scnerf_npp_composite_fwd")
        ctx.dims = (n, sf, sb, ray_o.shape, fg_z_max.shape, fg_z.shape)
        ctx.nets = (fg_net, bg_net)
        ctx.train = train
        if train:
            ctx.state = (o, d, zmax, zf, zb, fg_pts, bg_pts, views, raw_f, raw_b, save_f, save_b,
                         ops.pack_weights(flat_f, "bwd", pd=3, remap=fg_net.pack_remap()),
                         ops.pack_weights(flat_b, "bwd", pd=4, remap=bg_net.pack_remap()), pl_f, pl_b, mx_f, mx_b)
        tails = {"rgb": (3,), "fg_weights": (sf,), "bg_weights": (sb,), "fg_rgb": (3,), "bg_rgb": (3,)}
        return tuple(t[k].view(*dots, *tails.get(k, ())) for k in _OUT_KEYS)

    @staticmethod
    def backward(ctx, *g):
        if not ctx.train:
            raise RuntimeError("NerfNet.forward was evaluated without gradient tracking")
        n, sf, sb, o_shape, zmax_shape, fgz_shape = ctx.dims
        fg_net, bg_net = ctx.nets
        (o, d, zmax, zf, zb, fg_pts, bg_pts, views, raw_f, raw_b, save_f, save_b, wb_f, wb_b, pl_f, pl_b, mx_f, mx_b) = ctx.state
        dev = o.device
        lib = _capi.load()
        gs = [None if x is None else x.reshape(n, -1).contiguous().float() for x in g]
        d_raw_f = torch.empty((n * sf, 4), dtype=torch.float32, device=dev)
        d_raw_b = torch.empty((n * sb, 4), dtype=torch.float32, device=dev)
        d_z = torch.empty((n, sf), dtype=torch.float32, device=dev)
        d_zmax = torch.empty(n, dtype=torch.float32, device=dev)
        d_norm = torch.empty(n, dtype=torch.float32, device=dev)
        _capi.check(lib.scnerf_npp_composite_bwd(_p(raw_f), _p(raw_b), _p(zf), _p(zmax), _p(zb), _p(d),
                                                 *[_p(x) for x in gs], _p(d_raw_f), _p(d_raw_b), _p(d_z), _p(d_zmax),
                                                 _p(d_norm), n, sf, sb, _stream()), "scnerf_npp_composite_bwd")
        # both networks' data gradients first, then both weight-gradient passes (one clock recovery after the bf16
        # weight-gradient GEMMs instead of two: functional.py)
        grads_f, d_pts_f, d_views_f = ops.mlp_bwd(d_raw_f, fg_pts, views, sf, wb_f, save_f, pd=3, planes=pl_f, maxima=mx_f)
        grads_b, d_pts_b, d_views_b = ops.mlp_bwd(d_raw_b, bg_pts, views, sb, wb_b, save_b, pd=4, planes=pl_b, maxima=mx_b)
        flat_gf = ops.nerf_wgrad(save_f, grads_f, d_raw_f, n * sf, pd=3, maxima=mx_f)
        flat_gb = ops.nerf_wgrad(save_b, grads_b, d_raw_b, n * sb, pd=4, maxima=mx_b)
        g_o, g_d = torch.empty_like(o), torch.empty_like(d)
        g_z = torch.empty((n, sf), dtype=torch.float32, device=dev)
        _capi.check(lib.scnerf_npp_points_bwd(_p(o), _p(d), _p(zf), _p(zb), _p(d_pts_f), _p(d_pts_b), _p(d_views_f),
                                              _p(d_views_b), _p(d_norm), _p(d_z), _p(g_o), _p(g_d), _p(g_z), n, sf, sb,
                                              _stream()), "scnerf_npp_points_bwd")
        ctx.state = None
        n_fg = len(list(fg_net.parameters()))
        need = ctx.needs_input_grad[8:]

        def per_param(net, flat, pd, wanted):
            # networks whose .grad tensors are views of one flat buffer (FusedAdam / FlatGradAllReduce) take the whole flat
            # gradient with ONE scatter-add (every target index occurs once: deterministic) -- returned tensor by tensor,
            # autograd accumulates them with 24 tiny launches per network and level
            into = net.attached_flat_grad() if all(wanted) else None
            if into is not None:
                into.index_add_(0, net.canonical_to_module_index(flat.device), flat)
                return [None] * len(wanted)
            lay = ML.layout(pd)
            by_canon = {name: flat[lay.param_offsets[name]: lay.param_offsets[name] + int(torch.Size(shape).numel())].view(shape)
                        for name, shape in lay.param_shapes}
            from .nerf_network import canonical_to_module_name
            by_module = {canonical_to_module_name(k): v for k, v in by_canon.items()}
            return [by_module[name] for name, _ in net.named_parameters()]

        return (g_o.view(o_shape), g_d.view(o_shape), d_zmax.view(zmax_shape), g_z.view(fgz_shape), None, None, None, None,
                *per_param(fg_net, flat_gf, 3, need[:n_fg]), *per_param(bg_net, flat_gb, 4, need[n_fg:]))


class NerfNet(nn.Module):
    """(ddp_model.py:48-143) foreground net on (x, y, z), background net on (x, y, z, 1/r)."""

    def __init__(self, args):
        super().__init__()
        self.fg_embedder_position = Embedder(input_dim=3, max_freq_log2=args.max_freq_log2 - 1,
                                             N_freqs=args.max_freq_log2)
        self.fg_embedder_viewdir = Embedder(input_dim=3, max_freq_log2=args.max_freq_log2_viewdirs - 1,
                                            N_freqs=args.max_freq_log2_viewdirs)
        self.fg_net = MLPNet(D=args.netdepth, W=args.netwidth, input_ch=self.fg_embedder_position.out_dim,
                             input_ch_viewdirs=self.fg_embedder_viewdir.out_dim, use_viewdirs=args.use_viewdirs)
        self.bg_embedder_position = Embedder(input_dim=4, max_freq_log2=args.max_freq_log2 - 1,
                                             N_freqs=args.max_freq_log2)
        self.bg_embedder_viewdir = Embedder(input_dim=3, max_freq_log2=args.max_freq_log2_viewdirs - 1,
                                            N_freqs=args.max_freq_log2_viewdirs)
        self.bg_net = MLPNet(D=args.netdepth, W=args.netwidth, input_ch=self.bg_embedder_position.out_dim,
                             input_ch_viewdirs=self.bg_embedder_viewdir.out_dim, use_viewdirs=args.use_viewdirs)

    def forward(self, ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals):
        """ray_o, ray_d [..., 3]; fg_z_max [...]; fg_z_vals, bg_z_vals [..., N] -> OrderedDict(rgb,
        fg_weights, bg_weights, fg_rgb, fg_depth, bg_rgb, bg_depth, bg_lambda) as the reference (:134-143).
        bg_z_vals receives no gradient (the inverse radii never depend on learnable quantities)."""
        self.fg_net.require_standard()
        self.bg_net.require_standard()
        if self.fg_net.pt_dims != 3 or self.bg_net.pt_dims != 4:
            raise NotImplementedError("foreground / background nets must take 3-D / 4-D points")
        if not _capi.on_device(ray_o):
            raise RuntimeError("NerfNet inputs must be on the GPU: scnerf_amd has no CPU path")
        fg_params = [p for _, p in self.fg_net.named_parameters()]
        bg_params = [p for _, p in self.bg_net.named_parameters()]
        outs = _NerfNetFunction.apply(ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals, self.fg_net, self.bg_net,
                                      torch.is_grad_enabled(), *fg_params, *bg_params)
        return OrderedDict(zip(_OUT_KEYS, outs))


def remap_name(name):
    """(ddp_model.py:146-154)"""
    name = name.replace('.', '-')
    if name[-1] == '/':
        name = name[:-1]
    idx = name.rfind('/')
    for i in range(2):
        if idx >= 0:
            idx = name[:idx].rfind('/')
    return name[idx + 1:]


class NerfNetWithAutoExpo(nn.Module):
    """(ddp_model.py:157-188) NerfNet + optional per-image exposure parameters (2 floats per image,
    torch-level as in the reference)."""

    def __init__(self, args, optim_autoexpo=False, img_names=None):
        super().__init__()
        self.nerf_net = NerfNet(args)
        self.optim_autoexpo = optim_autoexpo
        if self.optim_autoexpo:
            assert img_names is not None
            logger.info('Optimizing autoexposure!')
            self.img_names = [remap_name(x) for x in img_names]
            logger.info('\n'.join(self.img_names))
            self.autoexpo_params = nn.ParameterDict(
                OrderedDict([(x, nn.Parameter(torch.Tensor([0.5, 0.]))) for x in self.img_names]))

    def forward(self, ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals, img_name=None):
        ret = self.nerf_net(ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals)
        if img_name is not None:
            img_name = remap_name(img_name)
        if self.optim_autoexpo and (img_name in self.autoexpo_params):
            autoexpo = self.autoexpo_params[img_name]
            scale = torch.abs(autoexpo[0]) + 0.5
            shift = autoexpo[1]
            ret['autoexpo'] = (scale, shift)
        return ret
