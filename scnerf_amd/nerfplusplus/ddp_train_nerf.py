"""The per-ray helpers of the reference's NeRF++ training script (nerfplusplus/ddp_train_nerf.py:50-132)
on the HIP kernels, differentiable where the reference is."""
from __future__ import annotations

import torch

from .. import _capi
from ..ops import _f, _p, _stream

TINY_NUMBER = 1e-6


def _flat2(t, last):
    return t.reshape(-1, last).contiguous().float()


class _Intersect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ray_o, ray_d):
        o, d = _flat2(ray_o, 3), _flat2(ray_d, 3)
        _f(o, "ray_o"), _f(d, "ray_d")
        n = o.shape[0]
        far = torch.empty(n, dtype=torch.float32, device=o.device)
        flag = torch.zeros(1, dtype=torch.int32, device=o.device)
        _capi.check(_capi.load().scnerf_npp_intersect_fwd(_p(o), _p(d), _p(far), _p(flag), n, _stream()),
                    "scnerf_npp_intersect_fwd")
        ctx.save_for_backward(o, d)
        ctx.shape = ray_o.shape
        ctx.mark_non_differentiable(flag)
        return far.view(ray_o.shape[:-1]), flag

    @staticmethod
    def backward(ctx, g, _g_flag):
        o, d = ctx.saved_tensors
        n = o.shape[0]
        go, gd = torch.empty_like(o), torch.empty_like(d)
        g = g.reshape(-1).contiguous().float()
        _capi.check(_capi.load().scnerf_npp_intersect_bwd(_p(o), _p(d), _p(g), _p(go), _p(gd), n, _stream()),
                    "scnerf_npp_intersect_bwd")
        return go.view(ctx.shape), gd.view(ctx.shape)


def intersect_sphere(ray_o, ray_d, check=True):
    """(:50-68) depth of the point where each ray leaves the unit sphere.  `check` reproduces the
    reference's exception for cameras outside the sphere (one host sync; pass False in hot loops)."""
    far, flag = _Intersect.apply(ray_o, ray_d)
    if check:
        if int(flag.item()) != 0:
            raise Exception("\n        Not all your cameras are bounded by the unit sphere; please make sure \n"
                            "        the cameras are normalized properly!\n        ")
    return far


class _Perturb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_vals, t_rand):
        s = z_vals.shape[-1]
        z, t = _flat2(z_vals, s), _flat2(t_rand, s)
        out = torch.empty_like(z)
        _capi.check(_capi.load().scnerf_npp_perturb_fwd(_p(z), _p(t), _p(out), z.shape[0], s, _stream()),
                    "scnerf_npp_perturb_fwd")
        ctx.save_for_backward(t)
        ctx.shape = z_vals.shape
        return out.view(z_vals.shape)

    @staticmethod
    def backward(ctx, g):
        (t,) = ctx.saved_tensors
        s = ctx.shape[-1]
        g = _flat2(g, s)
        gz = torch.empty_like(g)
        _capi.check(_capi.load().scnerf_npp_perturb_bwd(_p(g), _p(t), _p(gz), g.shape[0], s, _stream()),
                    "scnerf_npp_perturb_bwd")
        return gz.view(ctx.shape), None


def perturb_samples(z_vals, _t_rand=None):
    """(:71-80) jitter every depth between the mid points to its neighbours; `_t_rand` (tests) injects the
    uniforms instead of drawing them."""
    t_rand = torch.rand_like(z_vals) if _t_rand is None else _t_rand
    return _Perturb.apply(z_vals, t_rand)


class _SamplePdf(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bins, weights, u):
        m = weights.shape[-1]
        ns = u.shape[-1]
        b, w, uu = _flat2(bins, m + 1), _flat2(weights, m), _flat2(u, ns)
        n = b.shape[0]
        samples = torch.empty((n, ns), dtype=torch.float32, device=b.device)
        need = ctx.needs_input_grad[0]
        ba = torch.empty((n, ns), dtype=torch.int32, device=b.device) if need else None
        t = torch.empty((n, ns), dtype=torch.float32, device=b.device) if need else None
        _capi.check(_capi.load().scnerf_npp_sample_pdf(_p(b), _p(w), _p(uu), _p(samples), _p(ba), _p(t), None, n, m, ns,
                                                       _stream()), "scnerf_npp_sample_pdf")
        if need:
            ctx.save_for_backward(ba, t)
        ctx.dims = (n, m, ns, bins.shape)
        return samples.view(*u.shape)

    @staticmethod
    def backward(ctx, g):
        ba, t = ctx.saved_tensors
        n, m, ns, shape = ctx.dims
        g = _flat2(g, ns)
        gb = torch.empty((n, m + 1), dtype=torch.float32, device=g.device)
        _capi.check(_capi.load().scnerf_npp_sample_pdf_bwd(_p(g), _p(ba), _p(t), _p(gb), n, m, ns, _stream()),
                    "scnerf_npp_sample_pdf_bwd")
        return gb.view(shape), None, None


def sample_pdf_state(bins, weights, u):
    """What the sampler kernel computed on its way, for the parity tests: (samples [n, ns], cdf [n, m + 1] -- the
    cumulated pdf with the leading 0 (:95-98) --, below [n, ns], above [n, ns] -- the comparison count of :113 and the
    index under it (int64)).  Same launch as sample_pdf."""
    m, ns = weights.shape[-1], u.shape[-1]
    b, w, uu = _flat2(bins, m + 1), _flat2(weights.detach(), m), _flat2(u, ns)
    n = b.shape[0]
    samples = torch.empty((n, ns), dtype=torch.float32, device=b.device)
    ba = torch.empty((n, ns), dtype=torch.int32, device=b.device)
    t = torch.empty((n, ns), dtype=torch.float32, device=b.device)
    cdf = torch.empty((n, m + 1), dtype=torch.float32, device=b.device)
    _capi.check(_capi.load().scnerf_npp_sample_pdf(_p(b), _p(w), _p(uu), _p(samples), _p(ba), _p(t), _p(cdf), n, m, ns,
                                                   _stream()), "scnerf_npp_sample_pdf")
    return samples, cdf, (ba & 0xffff).long(), (ba >> 16).long()


def sample_pdf(bins, weights, N_samples, det=False, _u=None):
    """(:83-132) bins [..., M+1], weights [..., M] -> samples [..., N_samples]; differentiable in `bins`
    (the reference's callers detach the weights, :457,465)."""
    dots_sh = list(weights.shape[:-1])
    if _u is not None:
        u = _u
    elif det:
        u = torch.linspace(0., 1., N_samples, device="cpu").to(bins.device)        # host linspace: CPU rounding of the knots
        u = u.view([1] * len(dots_sh) + [N_samples]).expand(dots_sh + [N_samples])
    else:
        u = torch.rand(*(dots_sh + [N_samples]), device=bins.device)
    return _SamplePdf.apply(bins, weights.detach(), u)


_IMAGE_KEYS = (("rgb", 3), ("fg_rgb", 3), ("fg_depth", 1), ("bg_rgb", 3), ("bg_depth", 1), ("bg_lambda", 1))


def render_single_image(rank, world_size, models, ray_sampler, chunk_size, camera_model, camera_idx=None):
    """(ddp_train_nerf.py:135-257) render every pixel of one image with the cascade of `models`, the rays
    split evenly over the `world_size` processes; rank 0 returns one OrderedDict per cascade level with
    [H, W(, 3)] CPU tensors (rgb, fg_rgb, fg_depth, bg_rgb, bg_depth, bg_lambda), the other ranks None.

    Same sampling as the reference (level 0: evenly spaced depths without jitter; later levels: the
    deterministic inverse-CDF samples merged by a sort).  MI355X differences: the per-chunk results stay on
    the GPU and are packed (12 floats per ray) so that ONE gather per level moves them (the reference does
    one CPU gather per key, after a `.cpu()` per key and chunk)."""
    from collections import OrderedDict
    import torch.distributed as dist
    with torch.no_grad():
        if camera_idx is not None:
            ray_batch = ray_sampler.get_all(camera_model, camera_idx, None, rank)
        else:
            ray_batch = ray_sampler.get_all(camera_model, None, ray_sampler, rank)
    n_pix = ray_batch['ray_d'].shape[0]
    if (n_pix // world_size) * world_size != n_pix:
        raise Exception('Number of pixels in the image is not divisible by the number of GPUs!\n\t# pixels: {}\n\t# GPUs: {}'
                        .format(n_pix, world_size))
    per = n_pix // world_size
    device = torch.device("cuda", rank) if isinstance(rank, int) else torch.device(rank)
    mine = {k: v[rank * per:(rank + 1) * per].to(device) for k, v in ray_batch.items() if torch.is_tensor(v)}
    levels = models['cascade_level']
    packed = [torch.empty((per, 12), dtype=torch.float32, device=device) for _ in range(levels)]
    with torch.no_grad():
        for s0 in range(0, per, chunk_size):
            sl = slice(s0, min(per, s0 + chunk_size))
            ray_o, ray_d, min_depth = mine['ray_o'][sl], mine['ray_d'][sl], mine['min_depth'][sl]
            dots_sh = list(ray_d.shape[:-1])
            ret = None
            for m in range(levels):
                net = models['net_{}'.format(m)]
                N_samples = models['cascade_samples'][m]
                if m == 0:
                    fg_far_depth = intersect_sphere(ray_o, ray_d)
                    step = (fg_far_depth - min_depth) / (N_samples - 1)
                    fg_depth = torch.stack([min_depth + i * step for i in range(N_samples)], dim=-1)
                    bg_depth = torch.linspace(0., 1., N_samples, device="cpu").view([1] * len(dots_sh) + [N_samples]) \
                        .expand(dots_sh + [N_samples]).to(device)
                else:
                    fg_mid = .5 * (fg_depth[..., 1:] + fg_depth[..., :-1])
                    fg_s = sample_pdf(bins=fg_mid, weights=ret['fg_weights'][..., 1:-1], N_samples=N_samples, det=True)
                    fg_depth, _ = torch.sort(torch.cat((fg_depth, fg_s), dim=-1))
                    bg_mid = .5 * (bg_depth[..., 1:] + bg_depth[..., :-1])
                    bg_s = sample_pdf(bins=bg_mid, weights=ret['bg_weights'][..., 1:-1], N_samples=N_samples, det=True)
                    bg_depth, _ = torch.sort(torch.cat((bg_depth, bg_s), dim=-1))
                ret = net(ray_o, ray_d, fg_far_depth, fg_depth, bg_depth)
                c = 0
                for key, w in _IMAGE_KEYS:
                    packed[m][sl, c:c + w] = ret[key].reshape(-1, w)
                    c += w
    out = None
    for m in range(levels):
        if world_size > 1:
            gathered = [torch.empty_like(packed[m]) for _ in range(world_size)] if rank == 0 else None
            if dist.get_backend() == "gloo":
                src = packed[m].cpu()
                gathered = [torch.empty_like(src) for _ in range(world_size)] if rank == 0 else None
                dist.gather(src, gathered, dst=0)
            else:
                dist.gather(packed[m], gathered, dst=0)
            full = torch.cat(gathered, 0) if rank == 0 else None
        else:
            full = packed[m]
        if rank == 0 or world_size == 1:
            if out is None:
                out = [OrderedDict() for _ in range(levels)]
            full = full.cpu()
            c = 0
            for key, w in _IMAGE_KEYS:
                out[m][key] = full[:, c:c + w].reshape((ray_sampler.H, ray_sampler.W, -1)).squeeze()
                c += w
    return out if (rank == 0 or world_size == 1) else None
