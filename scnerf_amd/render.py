"""MI355X-native drop-in for the reference's `render` module (/root/reference NeRF/render.py):
same function names, arguments, return structures and quirks, with the work done by the HIP
kernels behind include/scnerf_hip.h.

    render(...)            :18-141     render_path(...)   :143-183
    render_rays(...)       :186-300    raw2outputs(...)   :302-355
    ndc_rays / ndc_rays_camera :357-396
    batchify_rays(...)     :398-413    sample_pdf(...)    :417-460
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .functional import RawToOutputsFunction, RenderConfig, RenderRaysFunction, host_linspace
from .get_rays import (get_rays_full_image_no_camera, get_rays_full_image_use_camera,  # noqa: F401  (the
                       get_rays_kps_no_camera, get_rays_kps_use_camera, ndc_rays, ndc_rays_camera)  # reference's render module imports these too, render.py:9-14)
from .run_nerf_helpers import NeRF

to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)

# The reference prints a warning when an output holds NaN/Inf (:296-298), which costs 7 device->host
# synchronisations per chunk.  Off by default here; SCNERF_CHECK_NUMERICS=1 or verbose=True restores it.
CHECK_NUMERICS = os.environ.get("SCNERF_CHECK_NUMERICS", "0") == "1"


def _unwrap(net):
    """create_nerf wraps the networks in nn.DataParallel for checkpoint-key compatibility
    (reference create_nerf.py:56,64); the kernels want the module itself."""
    return net.module if isinstance(net, nn.DataParallel) else net


def _device_rand(shape, device, pytest):
    if pytest:                                    # the reference's fixed numpy stream (:252-255, :432-440)
        np.random.seed(0)
        return torch.Tensor(np.random.rand(*shape)).to(device)
    return torch.rand(shape, device=device)


def _host_side_draws(r, n, N_samples, N_importance, perturb, raw_noise_std, dev, pytest):
    """the draws of one render_rays call as the reference makes them, one tensor at a time (:252-257, :329-336, :425-440):
    injected values (`_randoms`, tests), the fixed numpy stream of `pytest` mode, else torch's device generator"""
    t_rand = u = noise_c = noise_f = None
    if perturb > 0.:
        t_rand = r["t_rand"] if "t_rand" in r else _device_rand((n, N_samples), dev, pytest)
    if raw_noise_std > 0.:
        if "noise_c" in r:
            noise_c = r["noise_c"]
        elif pytest:                              # uniform, not normal, in this mode (:333-336)
            np.random.seed(0)
            noise_c = torch.Tensor(np.random.rand(n, N_samples) * raw_noise_std).to(dev)
        else:
            noise_c = torch.randn((n, N_samples), device=dev) * raw_noise_std
    if N_importance > 0:
        if perturb > 0.:
            u = r["u"] if "u" in r else _device_rand((n, N_importance), dev, pytest)
        if raw_noise_std > 0.:
            tot = N_samples + N_importance
            if "noise_f" in r:
                noise_f = r["noise_f"]
            elif pytest:
                np.random.seed(0)
                noise_f = torch.Tensor(np.random.rand(n, tot) * raw_noise_std).to(dev)
            else:
                noise_f = torch.randn((n, tot), device=dev) * raw_noise_std
    return t_rand, u, noise_c, noise_f


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False,
                perturb=0., N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0.,
                verbose=False, pytest=False, _randoms=None):
    """Volumetric rendering of a ray batch; see the reference docstring (:199-231) for the
    arguments and the returned dict (rgb_map, disp_map, acc_map[, raw][, rgb0, disp0, acc0, z_std]).

    `network_query_fn`: the FusedNetworkQuery made by scnerf_amd.create_nerf (it carries the embedder configuration the
    fused kernels were built for) takes the fused path when the networks have the standard shape and the batch has view
    directions; any other callable -- or network shape -- is called as the reference calls it, between the same HIP
    samplers and compositing kernels (`_render_rays_opaque`).  `_randoms` (tests only) injects
    dict(t_rand=, u=, noise_c=, noise_f=) instead of drawing them."""
    net_c = _unwrap(network_fn)
    net_f = _unwrap(network_fine) if network_fine is not None else None
    if not ops._capi.on_device(ray_batch):
        raise RuntimeError("ray_batch must be on the GPU: scnerf_amd has no CPU path")
    # the fused path: the standard network(s) behind a query object that carries the standard encodings, view directions in
    # the batch.  Anything else at this boundary is an opaque callable, as in the reference (:186-300): called as it is.
    fused_for = getattr(network_query_fn, "fused_for", None)
    fused = (fused_for is not None and fused_for(net_c) and (net_f is None or fused_for(net_f)) and ray_batch.shape[-1] > 8)
    n = ray_batch.shape[0]
    dev = ray_batch.device
    r = _randoms or {}
    t_rand = u = noise_c = noise_f = None
    if not r and not pytest and (perturb > 0. or raw_noise_std > 0.):
        # the throughput path: every draw of this call from ONE launch (csrc/randoms.hip) instead of four generator launches
        # and their scaling passes; the `pytest` mode and injected draws take the branches below, as the reference's
        t_rand, u, noise_c, noise_f = ops.render_randoms(n, N_samples, N_importance, raw_noise_std, dev,
                                                         want_t_rand=perturb > 0., want_u=perturb > 0.)
    elif perturb > 0. or raw_noise_std > 0.:
        t_rand, u, noise_c, noise_f = _host_side_draws(r, n, N_samples, N_importance, perturb, raw_noise_std, dev, pytest)
    if not fused:
        return _render_rays_opaque(ray_batch, network_fn, network_query_fn, int(N_samples), retraw, bool(lindisp),
                                   int(N_importance), network_fine, bool(white_bkgd), t_rand, u, noise_c, noise_f,
                                   CHECK_NUMERICS or verbose)
    cfg = RenderConfig(int(N_samples), int(N_importance), bool(lindisp), bool(white_bkgd), torch.is_grad_enabled())
    params = list(net_c.ordered_parameters()) + (list(net_f.ordered_parameters()) if net_f is not None else [])
    (rgb_map, disp_map, acc_map, depth_map, raw, rgb0, disp0, acc0, depth0, z_std, z_vals,
     z_samples) = RenderRaysFunction.apply(ray_batch, cfg, t_rand, u, noise_c, noise_f, net_c, net_f, *params)

    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map}
    if retraw:
        ret['raw'] = raw
    if N_importance > 0:
        ret['rgb0'] = rgb0
        ret['disp0'] = disp0
        ret['acc0'] = acc0
        ret['z_std'] = z_std
    if CHECK_NUMERICS or verbose:
        for k in ret:
            if torch.isnan(ret[k]).any() or torch.isinf(ret[k]).any():
                print(f"! [Numerical Error] {k} contains nan or inf.")
    return ret


def _render_rays_opaque(ray_batch, network_fn, network_query_fn, N_samples, retraw, lindisp, N_importance, network_fine,
                        white_bkgd, t_rand, u, noise_c, noise_f, check):
    """render_rays with `network_query_fn(pts, viewdirs, network_fn)` as an opaque callable (reference :186-300): a
    network shape the fused kernels do not cover, other encoding widths, `use_viewdirs=False`, or a closure of the
    caller's own.  Only the network evaluation is the callable's; the per-ray work around it stays on the HIP kernels:
    stratified depths (coarse_sample), compositing forward / backward (RawToOutputsFunction), the cdf, the wave-ballot
    search, the inverse cdf and the rank merge (fine_sample) -- same kernels, hence the same sample indices, as the fused
    path.  The sample positions `o + z d` are the reference's own one-line tensor expression (differentiable in the rays)."""
    n = ray_batch.shape[0]
    dev = ray_batch.device
    rays = ray_batch.contiguous().float()
    rays_o, rays_d = rays[:, 0:3], rays[:, 3:6]
    viewdirs = rays[:, -3:] if rays.shape[-1] > 8 else None
    z_vals, _ = ops.coarse_sample(rays.detach(), host_linspace(N_samples, dev), None if t_rand is None else t_rand.contiguous().float(),
                                  lindisp)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    raw = network_query_fn(pts, viewdirs, network_fn)
    rgb_map, disp_map, acc_map, weights, _ = RawToOutputsFunction.apply(raw, z_vals, rays_d, noise_c, white_bkgd)
    ret = {}
    if N_importance > 0:
        rgb_map_0, disp_map_0, acc_map_0 = rgb_map, disp_map, acc_map
        if u is None:                                  # perturb == 0: the deterministic draw (:424-427)
            u = host_linspace(N_importance, dev)
        z_vals, _, z_samples, z_std, _, _ = ops.fine_sample(rays.detach(), z_vals, weights.detach().contiguous(),
                                                            u.contiguous().float())
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
        run_fn = network_fn if network_fine is None else network_fine
        raw = network_query_fn(pts, viewdirs, run_fn)
        rgb_map, disp_map, acc_map, weights, _ = RawToOutputsFunction.apply(raw, z_vals, rays_d, noise_f, white_bkgd)
        ret.update(rgb0=rgb_map_0, disp0=disp_map_0, acc0=acc_map_0, z_std=z_std)
    out = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map}
    if retraw:
        out['raw'] = raw
    out.update(ret)
    if check:
        for k in out:
            if torch.isnan(out[k]).any() or torch.isinf(out[k]).any():
                print(f"! [Numerical Error] {k} contains nan or inf.")
    return out


_COLOUR_KEYS = ("rgb0", "rgb1", "rgb_map")


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """Renders `rays_flat` [N, 8|11] in slices of `chunk` rays and joins the per-slice dicts
    (reference :398-413).  The colour outputs are saturated at 1 *in place* per slice, as there -- which
    also zeroes the gradient of saturated pixels.  `_randoms` (tests only): injected draws follow
    their rays."""
    randoms = kwargs.pop("_randoms", None)
    pieces = []
    for lo in range(0, rays_flat.shape[0], chunk):
        hi = lo + chunk
        if randoms is not None:
            kwargs["_randoms"] = {name: draw[lo:hi] for name, draw in randoms.items()}
        out = render_rays(rays_flat[lo:hi], **kwargs)
        for key in _COLOUR_KEYS:
            colour = out.get(key)
            if colour is not None:
                # in place like the reference's `ret[key][ret[key] >= 1.0] = 1.0`, but as a masked fill: boolean
                # index assignment runs nonzero() and blocks the host on the GPU twice per chunk
                colour.masked_fill_(colour >= 1.0, 1.0)
        pieces.append(out)
    if not pieces:
        return {}
    if len(pieces) == 1:
        return dict(pieces[0])
    return {key: torch.cat([piece[key] for piece in pieces], 0) for key in pieces[0]}


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, _noise=None):
    """raw [N,S,4|5], z_vals [N,S], rays_d [N,3] -> (rgb_map, disp_map, acc_map, weights, depth_map)
    (reference :302-355)."""
    noise = _noise
    if noise is None and raw_noise_std > 0.:
        shape = tuple(raw[..., 3].shape)
        if pytest:
            np.random.seed(0)
            noise = torch.Tensor(np.random.rand(*shape) * raw_noise_std).to(raw.device)
        else:
            noise = torch.randn(shape, device=raw.device) * raw_noise_std
    return RawToOutputsFunction.apply(raw, z_vals, rays_d, noise, bool(white_bkgd))


def sample_pdf(bins, weights, N_samples, det=False, pytest=False, _u=None):
    """Hierarchical sampling (reference :417-460); bins [N,B], weights [N,B-1] -> samples [N,N_samples].
    The samples are returned detached (render_rays detaches them anyway, :274)."""
    n = bins.shape[0]
    dev = bins.device
    if _u is not None:
        u = _u
    elif det:
        u = host_linspace(N_samples, dev)
    elif pytest:
        np.random.seed(0)
        u = torch.Tensor(np.random.rand(n, N_samples)).to(dev)
    else:
        u = torch.rand((n, N_samples), device=dev)
    samples, _, _ = ops.sample_pdf(bins.detach().contiguous().float(), weights.detach().contiguous().float(),
                                   u.contiguous().float())
    return samples


# ---- render(): where the rays of a call come from -------------------------------------------------------
# The reference's render() (:27-103) is a five-way if-chain over (rays given?, camera model?, mode); here
# each ray source is a table entry: the preconditions it asserts (argument name -> must be None / must be
# given) and a builder returning (rays_o, rays_d, focal-for-the-NDC-warp).  Failed preconditions raise
# AssertionError like the reference's asserts; a mode outside train / val / test without precomputed rays
# trips the same "should not appear" assertion.

def _rays_precomputed(a):
    rays_o, rays_d = a["rays"]
    return rays_o, rays_d, (a["noisy_focal"] if a["camera_model"] is None else None)


def _rays_trained_camera_train_view(a):
    i_map = np.asarray(a["i_map"])
    assert a["image_idx"] in i_map
    slot = np.where(i_map == a["image_idx"])[0][0]
    rays_o, rays_d = get_rays_full_image_use_camera(H=a["H"], W=a["W"], camera_model=a["camera_model"],
                                                    extrinsic=a["noisy_extrinsic"][slot])
    return rays_o, rays_d, None


def _rays_trained_camera_held_out_view(a):
    rays_o, rays_d = get_rays_full_image_use_camera(H=a["H"], W=a["W"], camera_model=a["camera_model"],
                                                    extrinsic=a["transform_align"])
    return rays_o, rays_d, None


def _rays_noisy_pinhole_train_view(a):
    focal = a["noisy_focal"]
    rays_o, rays_d = get_rays_full_image_no_camera(H=a["H"], W=a["W"], focal=focal,
                                                   extrinsic=a["noisy_extrinsic"][a["image_idx"]])
    return rays_o, rays_d, focal


def _rays_ground_truth_pinhole(a):
    focal = a["gt_intrinsic"][0][0].item()
    rays_o, rays_d = get_rays_full_image_no_camera(H=a["H"], W=a["W"], focal=focal,
                                                   extrinsic=a["gt_extrinsic"][a["image_idx"]])
    return rays_o, rays_d, focal


_GIVEN, _ABSENT = False, True          # value = "must be None"
# (camera model present, phase) -> (preconditions, builder); phase: "train" or "eval" (= val / test)
_RAY_SOURCES = {
    (True, "train"): ({"i_map": _GIVEN, "gt_intrinsic": _ABSENT, "gt_extrinsic": _ABSENT},
                      _rays_trained_camera_train_view),
    (True, "eval"): ({"noisy_focal": _ABSENT, "noisy_extrinsic": _ABSENT}, _rays_trained_camera_held_out_view),
    (False, "train"): ({"noisy_focal": _GIVEN, "noisy_extrinsic": _GIVEN}, _rays_noisy_pinhole_train_view),
    (False, "eval"): ({"gt_extrinsic": _GIVEN, "noisy_focal": _ABSENT, "noisy_extrinsic": _ABSENT},
                      _rays_ground_truth_pinhole),
}
_PHASE = {"train": "train", "val": "eval", "test": "eval"}


def _select_rays(a):
    if a["rays"] is not None:
        return _rays_precomputed(a)
    entry = _RAY_SOURCES.get((a["camera_model"] is not None, _PHASE.get(a["mode"])))
    assert entry is not None, "This message should not appear."
    preconditions, build = entry
    for name, must_be_none in preconditions.items():
        assert (a[name] is None) == must_be_none, \
            "render(mode=%r): argument %r must %sbe given" % (a["mode"], name, "not " if must_be_none else "")
    return build(a)


def render(H, W, chunk, rays=None, noisy_focal=None, noisy_extrinsic=None, ndc=True, near=0., far=1.,
           use_viewdirs=False, mode=None, camera_model=None, image_idx=None, i_map=None, gt_intrinsic=None,
           gt_extrinsic=None, transform_align=None, _ray_range=None, **kwargs):
    """Ray-source selection, view directions, NDC warp, ray-batch packing and output reshaping of the
    reference's render() (:18-141); returns [rgb_map, disp_map, acc_map, extras].  `_ray_range=(lo, hi)`
    (not in the reference; used by the multi-GPU image renderer) renders only the flattened rays [lo, hi):
    outputs come back as [hi-lo, ...]."""
    assert mode is not None
    rays_o, rays_d, focal = _select_rays(dict(
        H=H, W=W, rays=rays, noisy_focal=noisy_focal, noisy_extrinsic=noisy_extrinsic, mode=mode,
        camera_model=camera_model, image_idx=image_idx, i_map=i_map, gt_intrinsic=gt_intrinsic,
        gt_extrinsic=gt_extrinsic, transform_align=transform_align))
    lead = tuple(rays_d.shape[:-1])                              # outputs are reshaped back to this
    # unit view directions of the UN-warped rays (:105-109), NDC warp (:113-117), near / far columns and the
    # concatenation (:119-128): one fused launch (csrc/camera_rays.hip pack_rays), differentiable w.r.t. the rays
    # and, through the camera model's focal lengths, its intrinsics
    from .camera_functional import pack_ray_batch
    packed = pack_ray_batch(H, W, rays_o, rays_d, near, far, use_viewdirs, ndc, focal=focal, camera_model=camera_model)
    if _ray_range is not None:
        packed = packed[_ray_range[0]:_ray_range[1]]
        lead = (packed.shape[0],)
    outputs = batchify_rays(packed, chunk, **kwargs)
    outputs = {key: value.reshape(lead + tuple(value.shape[1:])) for key, value in outputs.items()}
    maps = [outputs.pop(key) for key in ('rgb_map', 'disp_map', 'acc_map')]
    return maps + [outputs]


class _HostRing:
    """Pinned host staging for finished images: the device->host copy of image i runs on a side
    stream while image i+1 renders; nothing blocks the host until the pixels are needed."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.pending = []

    def push(self, *tensors):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.stream.wait_event(ev)
        hosts = []
        with torch.cuda.stream(self.stream):
            for t in tensors:
                h = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
                h.copy_(t, non_blocking=True)
                t.record_stream(self.stream)
                hosts.append(h)
        done = torch.cuda.Event()
        done.record(self.stream)
        self.pending.append((done, hosts))
        return len(self.pending) - 1

    def get(self, i):
        done, hosts = self.pending[i]
        done.synchronize()
        return [h.numpy() for h in hosts]


def _image_kwargs(i, hwf, chunk, render_kwargs, mode, camera_model, noisy_extrinsic, gt_intrinsic, gt_extrinsic,
                  i_map, transform_align):
    H, W, noisy_focal = hwf
    return dict(H=H, W=W, noisy_focal=noisy_focal, chunk=chunk, noisy_extrinsic=noisy_extrinsic,
                gt_intrinsic=gt_intrinsic, gt_extrinsic=gt_extrinsic, mode=mode, camera_model=camera_model,
                image_idx=i_map[i] if not i_map is None else i, i_map=i_map,
                transform_align=transform_align[i] if transform_align is not None else None, **render_kwargs)


def render_path(render_poses, hwf, chunk, render_kwargs, mode, gt_imgs=None, args=None, savedir=None,
                camera_model=None, noisy_extrinsic=None, gt_intrinsic=None, gt_extrinsic=None, i_map=None,
                transform_align=None):
    """Full-image rendering loop of the reference (:143-183): one image per entry of `render_poses`,
    returns (rgbs [n,H,W,3], disps [n,H,W]) as numpy; PNGs are written when `savedir` is given.
    Forward only (no activation workspaces); finished images leave the GPU through pinned buffers on a
    copy stream, so rendering image i+1 overlaps the transfer (and PNG encoding) of image i."""
    H, W, _ = hwf
    from .get_rays import KEYPOINT_CHECK
    KEYPOINT_CHECK.flush()          # an evaluation boundary: report any out-of-range key-point batch of the training steps
    ring = None
    rgbs, disps = [], []
    try:
        import tqdm
        it = tqdm.tqdm(render_poses)
    except Exception:
        it = render_poses

    def collect(j):
        rgb, disp = ring.get(j)
        rgbs.append(rgb)
        disps.append(disp)
        if savedir is not None:
            import imageio
            imageio.imwrite(os.path.join(savedir, '{:03d}.png'.format(j)), to8b(rgb))

    for i, _extrinsic in enumerate(it):
        with torch.no_grad():
            rgb, disp, _acc, _ = render(**_image_kwargs(i, hwf, chunk, render_kwargs, mode, camera_model,
                                                        noisy_extrinsic, gt_intrinsic, gt_extrinsic, i_map,
                                                        transform_align))
        if ring is None:
            ring = _HostRing(rgb.device)
        ring.push(rgb.reshape((H, W, 3)), disp.reshape((H, W)))
        if i > 0:
            collect(i - 1)                       # image i is rendering / copying meanwhile
    if ring is not None:
        collect(len(ring.pending) - 1)
    return np.stack(rgbs, 0), np.stack(disps, 0)
