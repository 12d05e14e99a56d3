"""Typed wrappers over the C ABI (include/scnerf_hip.h): torch CUDA(ROCm) tensors in,
kernels enqueued on torch's current stream.  No computation happens in Python and there
is no CPU fallback: a missing library or a CPU tensor raises."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _capi
from . import mlp_layout as ML

Tensor = torch.Tensor


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t: Tensor, dtype, name):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (scnerf_amd has no CPU path)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def _f(t, name):
    return _chk(t, torch.float32, name)


def searchsorted(a: Tensor, v: Tensor, side: str = "right") -> Tensor:
    """Batched search with broadcast rows, int64 result (torchsearchsorted.searchsorted
    semantics, NeRF/torchsearchsorted/src/torchsearchsorted/searchsorted.py:20-53)."""
    _f(a, "a"), _f(v, "v")
    nrow = max(a.shape[0], v.shape[0])
    out = torch.empty((nrow, v.shape[1]), dtype=torch.int64, device=a.device)
    st = _capi.load().scnerf_searchsorted(_p(a), _p(v), _p(out), nrow, a.shape[0], v.shape[0],
                                          a.shape[1], v.shape[1], int(side == "left"), _stream())
    _capi.check(st, "scnerf_searchsorted")
    return out


def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor, want_inds=False, want_cdf=False):
    _f(bins, "bins"), _f(weights, "weights"), _f(u, "u")
    n, nb = bins.shape
    ns = u.shape[-1]
    stride = ns if u.dim() == 2 else 0
    samples = torch.empty((n, ns), dtype=torch.float32, device=bins.device)
    inds = torch.empty((n, ns), dtype=torch.int64, device=bins.device) if want_inds else None
    cdf = torch.empty((n, nb), dtype=torch.float32, device=bins.device) if want_cdf else None
    st = _capi.load().scnerf_sample_pdf(_p(bins), _p(weights), _p(u), stride, _p(samples), _p(inds),
                                        _p(cdf), n, nb, ns, _stream())
    _capi.check(st, "scnerf_sample_pdf")
    return samples, inds, cdf


def coarse_sample(rays: Tensor, t_vals: Tensor, t_rand: Optional[Tensor], lindisp: bool):
    _f(rays, "rays"), _f(t_vals, "t_vals")
    n, s = rays.shape[0], t_vals.shape[0]
    if t_rand is not None:
        _f(t_rand, "t_rand")
    z = torch.empty((n, s), dtype=torch.float32, device=rays.device)
    pts = torch.empty((n, s, 3), dtype=torch.float32, device=rays.device)
    st = _capi.load().scnerf_coarse_sample(_p(rays), rays.shape[1], _p(t_vals), _p(t_rand), _p(z),
                                           _p(pts), n, s, int(bool(lindisp)), _stream())
    _capi.check(st, "scnerf_coarse_sample")
    return z, pts


def fine_sample(rays: Tensor, z_c: Tensor, w_c: Tensor, u: Tensor, want_inds=False, want_cdf=False):
    _f(rays, "rays"), _f(z_c, "z_c"), _f(w_c, "w_c"), _f(u, "u")
    n, sc = z_c.shape
    sf = u.shape[-1]
    stride = sf if u.dim() == 2 else 0
    dev = rays.device
    z_f = torch.empty((n, sc + sf), dtype=torch.float32, device=dev)
    pts_f = torch.empty((n, sc + sf, 3), dtype=torch.float32, device=dev)
    z_s = torch.empty((n, sf), dtype=torch.float32, device=dev)
    z_std = torch.empty((n,), dtype=torch.float32, device=dev)
    inds = torch.empty((n, sf), dtype=torch.int64, device=dev) if want_inds else None
    cdf = torch.empty((n, sc - 1), dtype=torch.float32, device=dev) if want_cdf else None
    st = _capi.load().scnerf_fine_sample(_p(rays), rays.shape[1], _p(z_c), _p(w_c), _p(u), stride,
                                         _p(z_f), _p(pts_f), _p(z_s), _p(z_std), _p(inds), _p(cdf),
                                         n, sc, sf, _stream())
    _capi.check(st, "scnerf_fine_sample")
    return z_f, pts_f, z_s, z_std, inds, cdf


_index_cache = {}


def _device_index(kind: str, device) -> Tensor:
    key = (kind, str(device))
    if key not in _index_cache:
        idx = ML.forward_index() if kind == "fwd" else ML.backward_index()
        _index_cache[key] = torch.from_numpy(idx).to(device)
    return _index_cache[key]


def check_layout():
    out = (np.zeros(32, np.int32))
    import ctypes
    st = _capi.load().scnerf_mlp_layout_info(out.ctypes.data_as(ctypes.c_void_p), 32)
    _capi.check(st, "scnerf_mlp_layout_info")
    exp = [ML.FWD_STREAM, ML.FWD_BIAS, ML.FWD_BIAS_F, ML.FWD_BIAS_V, ML.FWD_BIAS_RGB, ML.FWD_ALPHA_W,
           ML.FWD_ALPHA_B, ML.FWD_TOTAL, ML.BWD_STREAM, ML.BWD_ALPHA_W, ML.BWD_TOTAL,
           ML.SAVE_FLOATS_PER_SAMPLE, ML.GRAD_FLOATS_PER_SAMPLE]
    if out[:len(exp)].tolist() != exp:
        raise RuntimeError("kernel / mlp_layout.py constants disagree: %s vs %s" % (out[:len(exp)].tolist(), exp))


def pack_weights(flat_params: Tensor, kind: str = "fwd", out: Optional[Tensor] = None) -> Tensor:
    """flat parameter buffer (reference order, 595 844 floats) -> packed streaming buffer."""
    _f(flat_params, "flat_params")
    if flat_params.numel() != ML.N_PARAMS:
        raise ValueError("expected %d parameters, got %d" % (ML.N_PARAMS, flat_params.numel()))
    idx = _device_index(kind, flat_params.device)
    if out is None:
        out = torch.empty(idx.numel(), dtype=torch.float32, device=flat_params.device)
    st = _capi.load().scnerf_gather_f32(_p(flat_params), _p(idx), _p(out), idx.numel(), _stream())
    _capi.check(st, "scnerf_gather_f32")
    return out


def mlp_fwd(pts: Tensor, viewdirs: Tensor, samples_per_ray: int, wpacked: Tensor,
            save: Optional[Tensor] = None) -> Tensor:
    _f(pts, "pts"), _f(viewdirs, "viewdirs"), _f(wpacked, "wpacked")
    P = pts.numel() // 3
    if wpacked.numel() != ML.FWD_TOTAL:
        raise ValueError("wpacked has the wrong size")
    if save is not None:
        _f(save, "save")
        if save.numel() < ML.save_floats(P):
            raise ValueError("activation workspace too small")
    raw = torch.empty((P, 4), dtype=torch.float32, device=pts.device)
    st = _capi.load().scnerf_mlp_fwd(_p(pts), _p(viewdirs), int(samples_per_ray), _p(wpacked), _p(raw),
                                     _p(save), P, _stream())
    _capi.check(st, "scnerf_mlp_fwd")
    return raw
