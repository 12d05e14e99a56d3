"""autograd glue between PyTorch and the HIP kernels (no arithmetic happens here).

`RenderRaysFunction` is one differentiable node for the whole of `render_rays`
(/root/reference NeRF/render.py:186-300): stratified sampling -> coarse network -> compositing
-> inverse-CDF sampling + merge -> fine network -> compositing, with gradients to the ray batch
(origin, direction, view direction) and to every parameter of both networks.
`RawToOutputsFunction` exposes compositing on its own (raw2outputs, :302-355).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import mlp_layout as ML
from . import ops

Tensor = torch.Tensor


@dataclass
class RenderConfig:
    n_samples: int
    n_importance: int
    lindisp: bool
    white_bkgd: bool
    # torch.is_grad_enabled() where render_rays was CALLED: inside Function.forward grad mode is always off, and
    # ctx.needs_input_grad stays True for parameters that require grad under torch.no_grad() -- without this flag a
    # forward-only call (render_path) would run the training instantiation and store every activation
    track: bool = True


_host_cache = {}


def host_linspace(n: int, device) -> Tensor:
    """torch.linspace(0, 1, n) computed on the HOST (the reference path under test is the CPU one
    and ATen's CPU / GPU linspace kernels may differ in the last bit; `device="cpu"` is explicit because the reference
    script makes CUDA the default tensor type, run_nerf.py:1046), cached per device."""
    key = (n, str(device))
    if key not in _host_cache:
        _host_cache[key] = torch.linspace(0.0, 1.0, steps=n, dtype=torch.float32, device="cpu").to(device)
    return _host_cache[key]


def _c(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    return t.contiguous().float()


class RenderRaysFunction(torch.autograd.Function):
    """apply(ray_batch, cfg, t_rand, u, noise_c, noise_f, net_c, net_f, *params_c, *params_f)

    t_rand [N,S_c] / None (perturb == 0); u [N,S_f] / None (deterministic linspace);
    noise_c/f: already scaled density noise or None.  `net_c` / `net_f` are the NeRF modules
    (flat parameter storage); their parameters are also passed as inputs so autograd routes the
    gradients.  Returns (rgb_map, disp_map, acc_map, depth_map, raw, rgb0, disp0, acc0, depth0,
    z_std, z_vals, z_samples); the *0 / z_std / z_samples entries are None when S_f == 0."""

    @staticmethod
    def forward(ctx, ray_batch, cfg, t_rand, u, noise_c, noise_f, net_c, net_f, *params):
        # outputs the loss does not touch arrive in backward as None (it handles that), not as ten zero tensors
        ctx.set_materialize_grads(False)
        rays = _c(ray_batch)
        if rays.shape[1] < 11:
            raise NotImplementedError("the fused network needs view directions (ray_batch [N, 11])")
        n = rays.shape[0]
        dev = rays.device
        sc, sf = cfg.n_samples, cfg.n_importance
        train = cfg.track and any(ctx.needs_input_grad)
        viewdirs = rays[:, 8:11]

        net_c.require_standard()
        flat_c = net_c.flat_parameters()
        save_c = ops.save_workspace(n * sc, dev) if train else None
        # the packed fp32 tables + what the arithmetic in force (ops.mlp_arithmetic) needs besides them: the resident
        # kernels' streams, or nothing (the fused fp32-MFMA kernels).  Training packs per call (the weights change every
        # step); a forward-only call takes them from the network's version-keyed cache (ops.inference_packs)
        if train or n == 0:
            wf_c = ops.pack_weights(flat_c, "fwd")
            pl_c = ops.pack_for_arithmetic(flat_c, train) if n > 0 else None
        else:
            wf_c, pl_c = ops.inference_packs(net_c, flat_c)
        resident = isinstance(pl_c, ops.ResidentWeights)
        # (resident kernels, training: they leave the chunk maxima the fp16 weight-gradient GEMMs scale by)
        mx_c = ops.ChunkMaxima(n * sc, dev) if (train and resident) else None
        if sc == ops.COARSE_STAGE_SAMPLES and n > 0:
            # the whole coarse stage -- stratified depths, network, compositing -- is one launch
            z_c, pts_c, raw_c, rgb_c, disp_c, acc_c, w_c, depth_c = ops.coarse_stage_fwd(
                rays, host_linspace(sc, dev), _c(t_rand), cfg.lindisp, wf_c, save_c, _c(noise_c), cfg.white_bkgd,
                planes=pl_c, maxima=mx_c)
        else:
            z_c, pts_c = ops.coarse_sample(rays, host_linspace(sc, dev), _c(t_rand), cfg.lindisp)
            raw_c = ops.mlp_fwd(pts_c, viewdirs, sc, wf_c, save_c, planes=pl_c, maxima=mx_c).view(n, sc, 4)
            rgb_c, disp_c, acc_c, w_c, depth_c = ops.composite_fwd(raw_c, z_c, rays, _c(noise_c), cfg.white_bkgd)

        ctx.cfg, ctx.train, ctx.n = cfg, train, n
        ctx.net_c, ctx.net_f = net_c, net_f
        ctx.n_params_c = len(net_c.ordered_parameters())
        ctx.coarse = (z_c, pts_c, raw_c, _c(noise_c), save_c, mx_c)
        ctx.pl_c = pl_c
        ctx.rays = rays
        ctx.wb_c = ops.pack_weights(flat_c, "bwd") if train else None
        ctx.fine = None
        if sf == 0:
            out = (rgb_c, disp_c, acc_c, depth_c, raw_c, None, None, None, None, None, z_c, None)
            ctx.mark_non_differentiable(z_c)
            return out

        fine_net = net_f if net_f is not None else net_c
        u_dev = _c(u) if u is not None else host_linspace(sf, dev)
        tot = sc + sf
        fine_net.require_standard()
        flat_f = fine_net.flat_parameters()
        save_f = ops.save_workspace(n * tot, dev) if train else None
        if fine_net is net_c:
            wf_f, pl_f = wf_c, pl_c
        elif train or n == 0:
            wf_f = ops.pack_weights(flat_f, "fwd")
            pl_f = ops.pack_for_arithmetic(flat_f, train) if n > 0 else None
        else:
            wf_f, pl_f = ops.inference_packs(fine_net, flat_f)
        mx_f = ops.ChunkMaxima(n * tot, dev) if (train and resident) else None
        if ops.fused_fine_stage() and resident and sc == ops.COARSE_STAGE_SAMPLES and sf in ops.FINE_STAGE_IMPORTANCE and n > 0:
            # the whole fine stage -- inverse-cdf sampler, merge, network, compositing -- as one launch (opt-in: measured
            # slower than the three launches below, ops.fused_fine_stage)
            z_f, pts_f, z_s, z_std, _, _, raw_f, rgb_f, disp_f, acc_f, depth_f, _ = ops.fine_stage_fwd(
                rays, z_c, w_c, u_dev, wf_f, save_f, _c(noise_f), cfg.white_bkgd, pl_f, maxima=mx_f)
        else:
            z_f, pts_f, z_s, z_std, _, _ = ops.fine_sample(rays, z_c, w_c, u_dev)
            raw_f = ops.mlp_fwd(pts_f, viewdirs, tot, wf_f, save_f, planes=pl_f, maxima=mx_f).view(n, tot, 4)
            rgb_f, disp_f, acc_f, _, depth_f = ops.composite_fwd(raw_f, z_f, rays, _c(noise_f), cfg.white_bkgd,
                                                                 want_weights=False)
        ctx.fine = (z_f, pts_f, raw_f, _c(noise_f), save_f, mx_f)
        ctx.pl_f = pl_f
        ctx.wb_f = (ctx.wb_c if fine_net is net_c else ops.pack_weights(flat_f, "bwd")) if train else None
        ctx.mark_non_differentiable(z_std, z_f, z_s)
        return (rgb_f, disp_f, acc_f, depth_f, raw_f, rgb_c, disp_c, acc_c, depth_c, z_std, z_f, z_s)

    @staticmethod
    def _stage_dgrad(stage, rays, spr, wbk, white_bkgd, g_rgb, g_disp, g_acc, g_depth, g_raw, d_rays, accumulate,
                     planes=None):
        """Data gradients of one stage (compositing, network, rays); -> what its weight gradients need."""
        z, pts, raw, noise, save, maxima = stage
        d_raw, d_rd = ops.composite_bwd(raw, z, rays, noise, white_bkgd, _c(g_rgb), _c(g_disp), _c(g_acc),
                                        _c(g_depth), _c(g_raw))
        grads, d_pts, d_views = ops.mlp_bwd(d_raw, pts, rays[:, 8:11], spr, wbk, save, planes=planes, maxima=maxima)
        ops.ray_reduce(d_pts, d_views, z, d_rd, d_rays, accumulate)
        return save, grads, d_raw, z.shape[0] * spr, maxima

    @staticmethod
    def _stage_wgrad(pending, into=None):
        """`into`: the network's attached flat .grad buffer -- the weight gradients are ADDED to it and
        None is returned (nothing for autograd to accumulate); otherwise a fresh flat gradient."""
        save, grads, d_raw, P, maxima = pending
        if into is not None:
            ops.nerf_wgrad(save, grads, d_raw, P, flat_grad=into, accumulate=True, maxima=maxima)
            return None
        return ops.nerf_wgrad(save, grads, d_raw, P, maxima=maxima)

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_depth, g_raw, g_rgb0, g_disp0, g_acc0, g_depth0, *_unused):
        if not ctx.train:
            raise RuntimeError("render_rays was evaluated without gradient tracking")
        cfg, rays, n = ctx.cfg, ctx.rays, ctx.n
        sc, sf = cfg.n_samples, cfg.n_importance
        d_rays = torch.zeros_like(rays)
        fg_c = fg_f = None
        wrote = False
        # networks whose .grad tensors are views of one flat buffer take their weight gradients by direct
        # accumulation (what AccumulateGrad would do with 24 returned tensors, in one launch)
        n_pc = ctx.n_params_c
        need = ctx.needs_input_grad[8:]
        into_c = ctx.net_c.attached_flat_grad() if all(need[:n_pc]) else None
        into_f = into_c if ctx.net_f is None else (ctx.net_f.attached_flat_grad() if all(need[n_pc:]) else None)
        # Both stages' data gradients first, then both stages' weight gradients: the 256 x 256 weight-gradient
        # GEMMs run on the bf16 matrix pipe at a lower shader clock, and the kernel that follows them inherits
        # that clock for ~0.3 ms -- adjacent, the two passes cost one such recovery per step instead of two.
        pend_f = pend_c = None
        if sf > 0:
            if any(g is not None for g in (g_rgb, g_disp, g_acc, g_depth, g_raw)):
                pend_f = RenderRaysFunction._stage_dgrad(ctx.fine, rays, sc + sf, ctx.wb_f, cfg.white_bkgd,
                                                         g_rgb, g_disp, g_acc, g_depth, g_raw, d_rays, wrote,
                                                         planes=ctx.pl_f)
                wrote = True
            coarse_g = (g_rgb0, g_disp0, g_acc0, g_depth0, None)
        else:
            coarse_g = (g_rgb, g_disp, g_acc, g_depth, g_raw)
        if any(g is not None for g in coarse_g):
            pend_c = RenderRaysFunction._stage_dgrad(ctx.coarse, rays, sc, ctx.wb_c, cfg.white_bkgd,
                                                     *coarse_g, d_rays, wrote, planes=ctx.pl_c)
        if pend_f is not None:
            fg_f = RenderRaysFunction._stage_wgrad(pend_f, into=into_f)
        if pend_c is not None:
            fg_c = RenderRaysFunction._stage_wgrad(pend_c, into=into_c)
        if sf > 0 and ctx.net_f is None and fg_f is not None:
            # one network serves both stages (reference :279): its gradient is the sum
            fg_c = fg_f if fg_c is None else fg_c + fg_f
            fg_f = None

        def split(flat):
            if flat is None:
                return [None] * len(ML.PARAM_SHAPES)
            return [flat[ML.PARAM_OFFSETS[name]: ML.PARAM_OFFSETS[name] + int(torch.Size(shape).numel())].view(shape)
                    for name, shape in ML.PARAM_SHAPES]

        grads = split(fg_c)
        if ctx.net_f is not None:
            grads += split(fg_f)
        ctx.coarse = ctx.fine = None       # release the activation workspaces
        return (d_rays, None, None, None, None, None, None, None, *grads)


class RawToOutputsFunction(torch.autograd.Function):
    """apply(raw, z_vals, rays_d, noise, white_bkgd) -> (rgb_map, disp_map, acc_map, weights, depth_map);
    differentiable in raw and rays_d (z_vals carry no gradient: see scnerf_composite_bwd)."""

    @staticmethod
    def forward(ctx, raw, z_vals, rays_d, noise, white_bkgd):
        raw4 = _c(raw[..., :4])            # a 5th channel, if any, is never read (reference :327,338)
        z = _c(z_vals)
        n = z.shape[0]
        rays = torch.zeros((n, 8), dtype=torch.float32, device=z.device)
        rays[:, 3:6] = rays_d
        noise = _c(noise)
        rgb, disp, acc, w, depth = ops.composite_fwd(raw4, z, rays, noise, white_bkgd)
        ctx.save_for_backward(raw4, z, rays, noise if noise is not None else torch.empty(0, device=z.device))
        ctx.white_bkgd = white_bkgd
        ctx.raw_channels = raw.shape[-1]
        ctx.mark_non_differentiable(w)
        return rgb, disp, acc, w, depth

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, _g_w, g_depth):
        raw4, z, rays, noise = ctx.saved_tensors
        noise = noise if noise.numel() else None
        d_raw, d_rd = ops.composite_bwd(raw4, z, rays, noise, ctx.white_bkgd, _c(g_rgb), _c(g_disp),
                                        _c(g_acc), _c(g_depth), None)
        if ctx.raw_channels > 4:
            pad = torch.zeros(d_raw.shape[:-1] + (ctx.raw_channels - 4,), device=d_raw.device)
            d_raw = torch.cat([d_raw, pad], -1)
        return d_raw, None, d_rd, None, None
