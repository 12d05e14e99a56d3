"""CPU oracle for the SCNeRF per-ray render path + camera ray generator.

TEST INFRASTRUCTURE ONLY.  This file is a CPU *restatement* of the reference's
algorithm (torch-CPU tensor ops, because that is where the reference's arithmetic
lives: SURVEY.md section 8c).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it -- and only as the checker, never
as the thing measured or shipped.  The product (`scnerf_amd`) never imports it.

Parity pinning: the reference's own tests do not pin this path,
so the oracle is pinned against outputs of the *unmodified reference itself*, run
in the build container by `oracle/gen_golden.py` and committed under
`tests/golden/` (see `tests/test_oracle_pinned.py`).

Differences from the reference in *form* (never in arithmetic):
  * randomness is injected (`t_rand`, `u`, `noise`) instead of drawn inside;
  * networks are plain dicts of tensors keyed by the reference's state-dict names;
  * every function works in the dtype of its inputs (float64 gives the "truth"
    run used as a yardstick for the fp32 noise floor).

All citations are relative to /root/reference.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- PE
def frequency_bands(n_freqs: int, dtype=torch.float32) -> Tensor:
    """2**linspace(0, L-1, L)  (NeRF/run_nerf_helpers.py:41, log_sampling=True)."""
    return 2.0 ** torch.linspace(0.0, float(n_freqs - 1), steps=n_freqs, dtype=dtype)


def positional_encoding(x: Tensor, n_freqs: int) -> Tensor:
    """[x, sin(f0 x), cos(f0 x), sin(f1 x), cos(f1 x), ...]; per frequency the 3
    sines precede the 3 cosines (NeRF/run_nerf_helpers.py:33-55).  n_freqs=10 ->
    63 channels, 4 -> 27.  n_freqs < 0 means identity (i_embed=-1, :58-59)."""
    if n_freqs < 0:
        return x
    parts = [x]
    for f in frequency_bands(n_freqs, x.dtype):
        xf = x * f
        parts.append(torch.sin(xf))
        parts.append(torch.cos(xf))
    return torch.cat(parts, dim=-1)


# -------------------------------------------------------------------------- MLP
def mlp_forward(p: Dict[str, Tensor], embedded: Tensor, input_ch: int,
                input_ch_views: int, skips=(4,), use_viewdirs=True, gates=None, record=None) -> Tensor:
    """NeRF.forward (NeRF/run_nerf_helpers.py:105-128): ReLU trunk with the encoded
    points re-concatenated *in front of* h after layer index in `skips`; density
    head from the trunk; colour head from [feature, encoded view dir].
    `gates` (tests only): depth + 1 boolean [P, width] tensors that REPLACE the ReLUs' own decisions (z * gate
    instead of relu(z)) -- a pre-activation within rounding of zero lands on either side in two fp32
    evaluations, and a comparison of gradients wants both sides on the same one.  `record`: a list that receives this
    evaluation's own decisions (z > 0), layer by layer."""
    x_pts, x_views = torch.split(embedded, [input_ch, input_ch_views], dim=-1)
    depth = len([k for k in p if k.startswith("pts_linears.") and k.endswith(".weight")])
    h = x_pts
    for i in range(depth):
        z = F.linear(h, p["pts_linears.%d.weight" % i], p["pts_linears.%d.bias" % i])
        if record is not None:
            record.append((z > 0).detach())
        h = F.relu(z) if gates is None else z * gates[i].to(z.dtype)
        if i in skips:
            h = torch.cat([x_pts, h], dim=-1)
    if use_viewdirs:
        sigma = F.linear(h, p["alpha_linear.weight"], p["alpha_linear.bias"])
        feat = F.linear(h, p["feature_linear.weight"], p["feature_linear.bias"])
        hv = torch.cat([feat, x_views], dim=-1)
        zv = F.linear(hv, p["views_linears.0.weight"], p["views_linears.0.bias"])
        if record is not None:
            record.append((zv > 0).detach())
        hv = F.relu(zv) if gates is None else zv * gates[depth].to(zv.dtype)
        rgb = F.linear(hv, p["rgb_linear.weight"], p["rgb_linear.bias"])
        return torch.cat([rgb, sigma], dim=-1)
    return F.linear(h, p["output_linear.weight"], p["output_linear.bias"])


def query_network(p, pts: Tensor, viewdirs: Optional[Tensor], multires=10,
                  multires_views=4, skips=(4,), gates=None, record=None) -> Tensor:
    """run_network (NeRF/create_nerf.py:18-32): encode points, broadcast + encode
    the per-ray view direction over the samples, concatenate, evaluate."""
    flat = pts.reshape(-1, pts.shape[-1])
    emb = positional_encoding(flat, multires)
    input_ch = emb.shape[-1]
    input_ch_views = 0
    if viewdirs is not None:
        dirs = viewdirs[:, None].expand(*pts.shape[:-1], viewdirs.shape[-1]).reshape(-1, viewdirs.shape[-1])
        emb_d = positional_encoding(dirs, multires_views)
        input_ch_views = emb_d.shape[-1]
        emb = torch.cat([emb, emb_d], dim=-1)
    out = mlp_forward(p, emb, input_ch, input_ch_views, skips, viewdirs is not None, gates, record)
    return out.reshape(list(pts.shape[:-1]) + [out.shape[-1]])


# ---------------------------------------------------------------- compositing
def composite(raw: Tensor, z_vals: Tensor, rays_d: Tensor, noise=None,
              white_bkgd=False):
    """raw2outputs (NeRF/render.py:302-355).  `noise` is the already-scaled additive
    density noise ([N,S]) or None.  Returns rgb_map, disp_map, acc_map, weights,
    depth_map."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    far_pad = torch.full_like(dists[..., :1], 1e10)                     # :319-323
    dists = torch.cat([dists, far_pad], dim=-1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)            # :325
    rgb = torch.sigmoid(raw[..., :3])                                   # :327
    dens = raw[..., 3] if noise is None else raw[..., 3] + noise
    alpha = 1.0 - torch.exp(-F.relu(dens) * dists)                      # :316-317
    ones = torch.ones((alpha.shape[0], 1), dtype=alpha.dtype)
    trans = torch.cumprod(torch.cat([ones, 1.0 - alpha + 1e-10], dim=-1), dim=-1)[:, :-1]
    weights = alpha * trans                                             # :340-344
    rgb_map = torch.sum(weights[..., None] * rgb, dim=-2)
    depth_map = torch.sum(weights * z_vals, dim=-1)
    acc_map = torch.sum(weights, dim=-1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map),
                               depth_map / (acc_map + 1e-10))           # :348-349
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


# ---------------------------------------------------------- inverse-CDF sampling
def aten_rowsum_f32(w: Tensor) -> Tensor:
    """Explicit restatement of what `torch.sum(w, -1)` computes for a contiguous
    float32 [N, M] tensor, 8 <= M < 512, with torch 2.10's CPU kernel (ATen
    SumKernel.cpp: vectorized_inner_sum -> row_sum -> multi_row_sum; the sum stub is
    built for 8-lane vectors even on AVX512 hosts -- verified here: the 8-lane order
    reproduces torch.sum bit-for-bit on this AVX512 container, the 16-lane one does
    not).  With nv = M // 8 full vectors v_0..v_{nv-1}:
      four interleaved vector accumulators p_k = sum_i v_{4i+k}  (i < nv // 4),
      left-over vectors (index >= 4*(nv//4)) are added to p_0 in order,
      then p_0 += p_1, p_2, p_3;
      the scalar tail (elements >= 8*nv) is summed sequentially from 0,
      and the 8 lanes of p_0 are added to it in lane order.
    This makes the pdf normaliser of sample_pdf host-independent."""
    assert w.dtype == torch.float32 and w.dim() == 2
    n, m = w.shape
    V, ILP = 8, 4
    nv = m // V
    assert 1 <= nv and nv // ILP < 16, "restatement covers the single-level cascade only"
    vec = [w[:, V * i:V * i + V] for i in range(nv)]
    p = [torch.zeros(n, V, dtype=torch.float32) for _ in range(ILP)]
    for i in range(nv // ILP):
        for k in range(ILP):
            p[k] = p[k] + vec[i * ILP + k]
    for i in range((nv // ILP) * ILP, nv):
        p[0] = p[0] + vec[i]
    for k in range(1, ILP):
        p[0] = p[0] + p[k]
    acc = torch.zeros(n, dtype=torch.float32)
    for k in range(nv * V, m):
        acc = acc + w[:, k]
    for k in range(V):
        acc = acc + p[0][:, k]
    return acc


def pdf_cdf(weights: Tensor, rowsum="torch") -> Tensor:
    """cdf = [0, cumsum((w+1e-5)/sum(w+1e-5))]  (NeRF/render.py:419-422)."""
    w = weights + 1e-5
    if rowsum == "torch":
        tot = torch.sum(w, dim=-1, keepdim=True)
    elif rowsum == "aten":
        tot = aten_rowsum_f32(w.contiguous())[:, None]
    else:
        raise ValueError(rowsum)
    pdf = w / tot
    cdf = torch.cumsum(pdf, dim=-1)
    return torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)


def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor, rowsum="torch"):
    """Hierarchical sampling (NeRF/render.py:417-460) with the uniform variates `u`
    ([N, S_f]) injected.  Returns (samples, inds, cdf); `inds` is the upper-bound
    (side='right') index into cdf (:444)."""
    cdf = pdf_cdf(weights, rowsum)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    c0 = torch.gather(cdf, 1, below)
    c1 = torch.gather(cdf, 1, above)
    b0 = torch.gather(bins, 1, below)
    b1 = torch.gather(bins, 1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)     # :455-456
    t = (u - c0) / denom
    samples = b0 + t * (b1 - b0)
    return samples, inds, cdf


def deterministic_u(n_rays: int, n_samples: int, dtype=torch.float32) -> Tensor:
    """u for perturb == 0 (det=True): linspace(0,1,S_f) per ray (render.py:425-427)."""
    return torch.linspace(0.0, 1.0, steps=n_samples, dtype=dtype).expand(n_rays, n_samples)


# ---------------------------------------------------------- stratified sampling
def stratified_z(near: Tensor, far: Tensor, n_samples: int, lindisp=False,
                 t_rand: Optional[Tensor] = None) -> Tensor:
    """Coarse depths (NeRF/render.py:235-257).  near/far: [N,1].  `t_rand` [N,S] in
    [0,1) jitters each sample inside its stratum; None = bin centres untouched
    (perturb == 0)."""
    t = torch.linspace(0.0, 1.0, steps=n_samples, dtype=near.dtype)
    if not lindisp:
        z = near * (1.0 - t) + far * t
    else:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    z = z.expand(near.shape[0], n_samples)
    if t_rand is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], dim=-1)
        lower = torch.cat([z[..., :1], mids], dim=-1)
        z = lower + (upper - lower) * t_rand
    return z


# ------------------------------------------------------------------ render_rays
def render_rays(ray_batch: Tensor, coarse, fine, n_samples: int, n_importance: int,
                t_rand=None, u=None, noise_c=None, noise_f=None, lindisp=False,
                white_bkgd=False, multires=10, multires_views=4, rowsum="torch",
                retraw=True, skips=(4,), z_samples=None, gates_coarse=None, gates_fine=None, record_gates=None):
    """render_rays (NeRF/render.py:186-300) with injected randomness.

    ray_batch [N, 8 or 11] = [o(3), d(3), near, far, (viewdirs(3))].  `fine` may be
    None (then the coarse net is re-used: :279).  When n_importance > 0 and `u` is
    None the deterministic linspace is used (perturb == 0).  Extra keys beyond the
    reference's dict (depth maps, z values, indices) are returned for the tests.
    `z_samples` [N, n_importance]: use THESE new depths instead of the sampler's own (they are detached anyway, :274) --
    for tests that compare what lies behind the sampler on identical samples; `gates_coarse` / `gates_fine`: mlp_forward's
    `gates` for the two network evaluations; `record_gates`: a dict that receives their own decisions ("coarse" / "fine").
    """
    n = ray_batch.shape[0]
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    z_c = stratified_z(near, far, n_samples, lindisp, t_rand)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_c[:, :, None]
    rec_c = record_gates.setdefault("coarse", []) if record_gates is not None else None
    rec_f = record_gates.setdefault("fine", []) if record_gates is not None else None
    raw_c = query_network(coarse, pts, viewdirs, multires, multires_views, skips, gates_coarse, rec_c)
    rgb, disp, acc, w, depth = composite(raw_c, z_c, rays_d, noise_c, white_bkgd)
    out = {"z_coarse": z_c, "weights_coarse": w}
    raw = raw_c
    if n_importance > 0:
        out.update(rgb0=rgb, disp0=disp, acc0=acc, depth0=depth, raw0=raw_c)
        z_mid = 0.5 * (z_c[..., 1:] + z_c[..., :-1])
        if u is None:
            u = deterministic_u(n, n_importance, z_c.dtype)
        z_s, inds, cdf = sample_pdf(z_mid, w[..., 1:-1], u, rowsum)
        z_s = z_s.detach()                                              # :274
        if z_samples is not None:
            z_s = z_samples.to(z_c.dtype)
        z_f, _ = torch.sort(torch.cat([z_c, z_s], dim=-1), dim=-1)
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z_f[:, :, None]
        raw = query_network(coarse if fine is None else fine, pts, viewdirs,
                            multires, multires_views, skips, gates_fine, rec_f)
        rgb, disp, acc, w, depth = composite(raw, z_f, rays_d, noise_f, white_bkgd)
        out.update(z_samples=z_s, inds=inds, cdf=cdf, z_fine=z_f, weights_fine=w,
                   z_std=torch.std(z_s, dim=-1, unbiased=False))         # :294
    out.update(rgb_map=rgb, disp_map=disp, acc_map=acc, depth_map=depth)
    if retraw:
        out["raw"] = raw
    return out


def clamp_rgb_inplace(ret: dict) -> dict:
    """batchify_rays' post-processing (NeRF/render.py:404-406): values >= 1 are
    overwritten with 1 in place, which also zeroes their gradient."""
    for key in ("rgb0", "rgb1", "rgb_map"):
        if key in ret:
            ret[key] = torch.where(ret[key] >= 1.0, torch.ones_like(ret[key]), ret[key])
    return ret


# --------------------------------------------------------------------------- NDC
def ndc_rays(H, W, fx, fy, near, rays_o, rays_d):
    """ndc_rays / ndc_rays_camera (NeRF/render.py:357-396); fx == fy == focal gives
    the first, the camera model's K[0][0], K[1][1] the second."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    sx = -1.0 / (W / (2.0 * fx))
    sy = -1.0 / (H / (2.0 * fy))
    o0 = sx * rays_o[..., 0] / rays_o[..., 2]
    o1 = sy * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1.0 + 2.0 * near / rays_o[..., 2]
    d0 = sx * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = sy * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2.0 * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


# ------------------------------------------------------------------------ camera
def _unit(v):
    mag = torch.sqrt((v ** 2).sum(1, keepdim=True))
    return v / (torch.clamp(mag, min=1e-8) + 1e-10)


def ortho6d_to_rotation(p6: Tensor) -> Tensor:
    """Gram-Schmidt of two 3-vectors into a rotation whose *columns* are x, y, x×y
    (model/camera_utils.py:78-133)."""
    a1, a2 = p6[:, 0:3], p6[:, 3:6]
    x = _unit(a1)
    dot = (x * a2).sum(1, keepdim=True)
    n2 = torch.clamp((x ** 2).sum(1, keepdim=True), min=1e-8)
    y = _unit(a2 - dot / (n2 + 1e-10) * x)
    z = torch.stack([x[:, 1] * y[:, 2] - x[:, 2] * y[:, 1],
                     x[:, 2] * y[:, 0] - x[:, 0] * y[:, 2],
                     x[:, 0] * y[:, 1] - x[:, 1] * y[:, 0]], dim=1)
    return torch.stack([x, y, z], dim=2)


def rotation_to_ortho6d(rot: Tensor) -> Tensor:
    """[C,3,3] -> [C,6]: the first two COLUMNS of each rotation side by side (model/camera_utils.py:136-137,
    `rotation2orth`): the inverse of ortho6d_to_rotation on orthonormal input, and how the camera model stores its
    initial poses (model/camera_model.py:137-141)."""
    return torch.cat([rot[:, :, 0], rot[:, :, 1]], dim=-1)


def camera_state(spec: dict, grad: bool = False, ray_d_from_ray_o: bool = False) -> Dict[str, Tensor]:
    """The oracle's camera dictionary from a synthetic camera spec (scnerf_amd.synthetic.camera_spec): initial
    intrinsics [fx, fy, cx, cy], initial extrinsics [6-D rotation | translation] (camera_model.py:130-141) and the
    learnable residuals (leaves when `grad`).  `ray_d_from_ray_o`: the Distortion model wraps ONE tensor in two
    Parameters (camera_model.py:224, :257-262) -- same values, two autograd leaves."""
    poses, K = spec["poses"], spec["K_init"]
    mk = (lambda x: x.clone().requires_grad_(True)) if grad else (lambda x: x.clone())
    return {"intrinsics_initial": torch.stack([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]),
            "extrinsics_initial": torch.cat([rotation_to_ortho6d(poses[:, :3, :3]), poses[:, :3, 3]], -1),
            "intrinsics_noise": mk(spec["intrinsics_noise"]), "extrinsics_noise": mk(spec["extrinsics_noise"]),
            "ray_o_noise": mk(spec["ray_o_noise"]),
            "ray_d_noise": mk(spec["ray_o_noise"] if ray_d_from_ray_o else spec["ray_d_noise"]),
            "intrinsics_noise_scale": spec["intrinsics_noise_scale"], "extrinsics_noise_scale": spec["extrinsics_noise_scale"],
            "ray_o_noise_scale": spec["ray_o_noise_scale"], "ray_d_noise_scale": spec["ray_d_noise_scale"],
            "multiplicative_noise": spec["multiplicative_noise"]}


def camera_intrinsic_params(cam: Dict[str, Tensor]) -> Tensor:
    """[fx, fy, cx, cy] after the learnable residual (model/camera_model.py:166-177)."""
    init, noise, s = cam["intrinsics_initial"], cam["intrinsics_noise"], cam["intrinsics_noise_scale"]
    if cam.get("multiplicative_noise", False):
        return init + noise * s * init
    return init + noise * s


def camera_extrinsics(cam: Dict[str, Tensor]):
    """Per-camera rotation [C,3,3] and translation [C,3] (camera_model.py:179-190)."""
    s = cam["extrinsics_noise_scale"]
    e = cam["extrinsics_initial"] + s * cam["extrinsics_noise"]
    return ortho6d_to_rotation(e[:, :6]), e[:, 6:]


def camera_K(cam: Dict[str, Tensor]) -> Tensor:
    """CameraModel.get_intrinsic(): the 4 x 4 matrix with fx, fy on the diagonal and cx, cy in column 2
    (model/camera_utils.py:191-195 on the parameters of model/camera_model.py:166-177)."""
    fx, fy, cx, cy = camera_intrinsic_params(cam)
    one, zero = torch.ones_like(fx), torch.zeros_like(fx)
    return torch.stack([torch.stack([fx, zero, cx, zero]), torch.stack([zero, fy, cy, zero]),
                        torch.stack([zero, zero, one, zero]), torch.stack([zero, zero, zero, one])])


def camera_E(cam: Dict[str, Tensor]) -> Tensor:
    """CameraModel.get_extrinsic(): [C, 4, 4] camera-to-world matrices, rotation from the 6-D parameters, translation in
    column 3 (model/camera_model.py:179-192, camera_utils.py:184-188)."""
    rot, trans = camera_extrinsics(cam)
    top = torch.cat([rot, trans[:, :, None]], -1)
    bottom = torch.zeros_like(top[:, :1, :])
    bottom = torch.cat([bottom[:, :, :3], torch.ones_like(bottom[:, :, :1])], -1)
    return torch.cat([top, bottom], 1)


def upsample_noise_grid(grid: Tensor, H: int, W: int, scale: float) -> Tensor:
    """CameraModel.get_ray_{o,d}_noise (camera_model.py:24-46): bilinear
    (align_corners=False) upsampling of the [H/g, W/g, 3] grid to [H*W, 3]."""
    up = F.interpolate(grid.permute(2, 0, 1)[None], (H, W), mode="bilinear",
                       align_corners=False)
    return up.permute(0, 2, 3, 1).reshape(-1, 3) * scale


def camera_rays(cam: Dict[str, Tensor], H: int, W: int, kps: Tensor, cam_idx: Tensor):
    """get_rays_kps_use_camera with per-ray camera indices (NeRF/get_rays.py:93-148).
    kps [N,2] float (x, y); cam_idx [N] int64."""
    fx, fy, cx, cy = camera_intrinsic_params(cam)
    K = torch.eye(3, dtype=fx.dtype)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, cx, cy
    Kinv = torch.inverse(K)
    R, t = camera_extrinsics(cam)
    hom = torch.stack([kps[:, 0], kps[:, 1], torch.ones_like(kps[:, 0])], dim=-1)
    dirs = hom @ Kinv.T
    dirs = torch.cat([dirs[:, :1], -dirs[:, 1:3]], dim=-1)               # :125
    Rn = R[cam_idx]
    rays_d = torch.sum(dirs[:, None, :] * Rn, dim=-1)
    rays_o = t[cam_idx]
    px = kps.long()                                                     # truncation (:135)
    lin = px[:, 1] * W + px[:, 0]
    if "ray_o_noise" in cam:
        rays_o = rays_o + upsample_noise_grid(cam["ray_o_noise"], H, W, cam["ray_o_noise_scale"])[lin]
    if "ray_d_noise" in cam:
        rays_d = rays_d + upsample_noise_grid(cam["ray_d_noise"], H, W, cam["ray_d_noise_scale"])[lin]
        rays_d = rays_d / (rays_d.norm(dim=1)[:, None] + 1e-10)
    return rays_o, rays_d


def pinhole_rays(H, W, focal, c2w: Tensor, kps: Tensor):
    """get_rays_kps_no_camera (NeRF/get_rays.py:75-90): integer pixel coordinates,
    dirs = [(x-W/2)/f, -(y-H/2)/f, -1] rotated by c2w[:3,:3], origin c2w[:3,3]."""
    px = kps.long()
    dirs = torch.stack([(px[:, 0] - W * 0.5) / focal, -(px[:, 1] - H * 0.5) / focal,
                        -torch.ones_like(px[:, 0])], dim=-1).to(c2w.dtype)
    rays_d = torch.sum(dirs[:, None, :] * c2w[:3, :3], dim=-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


# --------------------------------------------------------------------- optimizer
def adam_step(params, grads, exp_avgs, exp_avg_sqs, steps, lr, beta1=0.9, beta2=0.999, eps=1e-8,
              weight_decay=0.0, decay_idx_from=None):
    """One step of the reference's f_custom_adam (NeRF/create_nerf.py:199-254), amsgrad off: lists of
    tensors updated in place; `steps[i]` is the (already incremented) step count of tensor i; weight
    decay is added to the gradient of tensors with index >= decay_idx_from only (:219-226, :238-239).
    With decay_idx_from = len(params) this is torch.optim.Adam."""
    import math
    if decay_idx_from is None:
        decay_idx_from = len(params)
    for i, p in enumerate(params):
        g = grads[i]
        bc1 = 1 - beta1 ** steps[i]
        bc2 = 1 - beta2 ** steps[i]
        if weight_decay != 0 and i >= decay_idx_from:
            g = g.add(p, alpha=weight_decay)
        exp_avgs[i].mul_(beta1).add_(g, alpha=1 - beta1)
        exp_avg_sqs[i].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (exp_avg_sqs[i].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(exp_avgs[i], denom, value=-(lr / bc1))


def lr_schedule(lrate, lrate_decay, global_step):
    """run_nerf.py:617-621: lrate * 0.1 ** (global_step / (lrate_decay * 1000))."""
    return lrate * (0.1 ** (global_step / (lrate_decay * 1000)))


# ----------------------------------------------------------------------------- PRD loss

def prd_match_errors(kps0, kps1, rays0_o, rays0_d, rays1_o, rays1_d, K, E2, eps=1e-10, negate_fx=True):
    """Per match: squared re-projection errors (l0 in image 0, l1 in image 1) of the mutually closest points
    of the two rays, and the chirality flag t0, t1 > 0 (model/ray_dist_loss.py:97-208; the same arithmetic is
    filter_matches_with_gt, model/prd_evaluation.py:206-325, with eps = 1e-6)."""
    def unit(v):
        return v / (v.norm(p=2, dim=-1, keepdim=True) + eps)
    d0, d1 = unit(rays0_d), unit(rays1_d)
    w = rays0_o - rays1_o
    r = (d0 * d1).sum(-1)
    den = r ** 2 - 1 + eps
    t0 = ((d0 * w).sum(-1) - r * (d1 * w).sum(-1)) / den
    t1 = ((d1 * -w).sum(-1) - r * (d0 * -w).sum(-1)) / den
    p0 = t0[:, None] * d0 + rays0_o
    p1 = t1[:, None] * d1 + rays1_o
    Kk = K.clone()
    if negate_fx:
        Kk[0, 0] = -Kk[0, 0]

    def reproject(p, E):
        R, t = E[:3, :3], E[:3, 3]
        Rt = R.transpose(0, 1)
        q = p @ Rt.transpose(0, 1) + (-(Rt @ t))[None]         # R^T p - R^T t
        q4 = torch.cat([q, torch.ones_like(q[:, :1])], -1)
        n = q4 @ Kk.transpose(0, 1)
        return n[:, :2] / (n[:, 2:3] + eps)
    u01 = reproject(p0, E2[1])
    u10 = reproject(p1, E2[0])
    return ((u10 - kps0) ** 2).sum(-1), ((u01 - kps1) ** 2).sum(-1), (t0 > 0) & (t1 > 0)


def prd_loss(kps0, kps1, rays0_o, rays0_d, rays1_o, rays1_d, K, E2, threshold, eps=1e-10, negate_fx=True,
             eval_mode=False):
    """Projected-ray-distance loss of one image pair (model/ray_dist_loss.py:97-246), written per
    match instead of with batched einsums: the mutually closest points p0 / p1 of the two rays
    (:127-158), each re-projected into the other camera through E^-1 = [R^T | -R^T t] (:107-111,168-169)
    and K with K[0][0] negated for NeRF's axes (:102-105,170-176), squared pixel error against the
    matched key point (:203-208).  Train (:211-229): mean over {chirality t0,t1>0 and error < threshold
    and finite}, separately per direction, halved sum; also returns the count valid in both.
    Eval (:231-246): errors above the threshold / non-finite are set to it, mean over the
    chirality-valid.  Differentiable in every float argument (torch autograd)."""
    l0, l1, chir = prd_match_errors(kps0, kps1, rays0_o, rays0_d, rays1_o, rays1_d, K, E2, eps, negate_fx)
    l0, l1 = l0[chir], l1[chir]
    if not eval_mode:
        ok0 = (l0 < threshold) & torch.isfinite(l0)
        ok1 = (l1 < threshold) & torch.isfinite(l1)
        return 0.5 * (l0[ok0].mean() + l1[ok1].mean()), float((ok0 & ok1).sum())
    thr = torch.full_like(l0, threshold)
    l0 = torch.where((l0 > threshold) | ~torch.isfinite(l0), thr, l0)
    l1 = torch.where((l1 > threshold) | ~torch.isfinite(l1), thr, l1)
    return 0.5 * (l0.mean() + l1.mean()), None
