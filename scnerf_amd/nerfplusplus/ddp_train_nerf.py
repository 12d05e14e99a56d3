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
        _capi.check(_capi.load().scnerf_npp_sample_pdf(_p(b), _p(w), _p(uu), _p(samples), _p(ba), _p(t), n, m, ns,
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


def sample_pdf(bins, weights, N_samples, det=False, _u=None):
    """(:83-132) bins [..., M+1], weights [..., M] -> samples [..., N_samples]; differentiable in `bins`
    (the reference's callers detach the weights, :457,465)."""
    dots_sh = list(weights.shape[:-1])
    if _u is not None:
        u = _u
    elif det:
        u = torch.linspace(0., 1., N_samples).to(bins.device)        # host linspace: CPU rounding of the knots
        u = u.view([1] * len(dots_sh) + [N_samples]).expand(dots_sh + [N_samples])
    else:
        u = torch.rand(*(dots_sh + [N_samples]), device=bins.device)
    return _SamplePdf.apply(bins, weights.detach(), u)
