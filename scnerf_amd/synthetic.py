"""Deterministic synthetic inputs shaped like BASELINE.md section 2 (there are no datasets
in the build or GPU containers): rays, targets, injected randomness, network
weights with the reference initialisation, and a fern-like camera rig.

Pure torch-CPU generators (mt19937 streams are stable across hosts); callers move
the tensors to the device they need.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

Tensor = torch.Tensor


def xavier_nerf_params(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4,
                       skips=(4,), use_viewdirs=True, seed=0) -> Dict[str, Tensor]:
    """Weights drawn exactly like the reference's `NeRF(...)` constructed under
    torch.manual_seed(seed) (xavier_uniform, relu/linear gain, zero bias; creation
    order pts_linears, views_linears, feature, alpha, rgb:
    /root/reference NeRF/run_nerf_helpers.py:13-21, :88-103)."""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, Tensor] = {}

    def dense(name, fan_in, fan_out, act):
        gain = math.sqrt(2.0) if act == "relu" else 1.0
        bound = math.sqrt(3.0) * gain * math.sqrt(2.0 / float(fan_in + fan_out))
        p[name + ".weight"] = torch.empty(fan_out, fan_in).uniform_(-bound, bound, generator=g)
        p[name + ".bias"] = torch.zeros(fan_out)

    dense("pts_linears.0", input_ch, W, "relu")
    for i in range(D - 1):
        dense("pts_linears.%d" % (i + 1), W + input_ch if i in skips else W, W, "relu")
    dense("views_linears.0", input_ch_views + W, W // 2, "relu")
    if use_viewdirs:
        dense("feature_linear", W, W, "linear")
        dense("alpha_linear", W, 1, "linear")
        dense("rgb_linear", W // 2, 3, "linear")
    else:
        dense("output_linear", W, output_ch, "linear")
    return p


def network_params(seed=0, bias_scale=0.1, alpha_bias=0.5, **kw) -> Dict[str, Tensor]:
    """xavier weights + *non-zero* biases (so that bias handling is exercised) and a
    positive density bias (so that rays are neither empty nor fully opaque)."""
    p = xavier_nerf_params(seed=seed, **kw)
    g = torch.Generator().manual_seed(1000 + seed)
    for k in sorted(p):
        if k.endswith(".bias"):
            p[k] = (torch.rand(p[k].shape, generator=g) * 2 - 1) * bias_scale
    if "alpha_linear.bias" in p:
        p["alpha_linear.bias"] = p["alpha_linear.bias"] + alpha_bias
    return p


def ray_batch(n: int, seed=1, lindisp=False) -> Tensor:
    """[n, 11] = [o, d, near, far, viewdirs]: o ~ N(0, 0.1^2), d ~ N(0,1) with
    d_z <- -(|d_z| + 0.5) (forward-facing, NDC-like), near 0 / far 1."""
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(n, 3, generator=g) * 0.1
    d = torch.randn(n, 3, generator=g)
    d[:, 2] = -(d[:, 2].abs() + 0.5)
    near = torch.full((n, 1), 0.25 if lindisp else 0.0)
    far = torch.full((n, 1), 1.5 if lindisp else 1.0)
    v = d / torch.norm(d, dim=-1, keepdim=True)
    return torch.cat([o, d, near, far, v], dim=-1)


def target_rgb(n: int, seed=2) -> Tensor:
    return torch.rand(n, 3, generator=torch.Generator().manual_seed(seed))


def render_randoms(n: int, s_c: int, s_f: int, seed=3) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    out = {"t_rand": torch.rand(n, s_c, generator=g),
           "noise_c": torch.randn(n, s_c, generator=g)}
    if s_f > 0:
        out["u"] = torch.rand(n, s_f, generator=g)
        out["noise_f"] = torch.randn(n, s_c + s_f, generator=g)
    return out


def _rot(axis: Tensor, angle: float) -> Tensor:
    a = axis / axis.norm()
    K = torch.tensor([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return torch.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)


def camera_spec(H: int, W: int, n_cams=17, seed=4, multiplicative=True, grid_size=10,
                focal=400.0) -> Dict[str, object]:
    """A fern-like rig: n_cams poses within 30 degrees of identity, t ~ N(0, 0.3^2),
    K = [[f,0,W/2],[0,f,H/2]], N(0,1)-filled ray noise grids (scale 1e-3) and small
    non-zero intrinsic / extrinsic residuals so every learnable path is live."""
    g = torch.Generator().manual_seed(seed)
    poses = torch.zeros(n_cams, 4, 4)
    for c in range(n_cams):
        axis = torch.randn(3, generator=g)
        ang = float(torch.rand(1, generator=g)) * math.pi / 6
        poses[c, :3, :3] = _rot(axis, ang)
        poses[c, :3, 3] = torch.randn(3, generator=g) * 0.3
        poses[c, 3, 3] = 1.0
    K = torch.tensor([[focal, 0, W / 2, 0], [0, focal, H / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    gh, gw = H // grid_size, W // grid_size
    return {
        "K_init": K, "poses": poses,
        "intrinsics_noise": torch.randn(4, generator=g) * 1e-2,
        "extrinsics_noise": torch.randn(n_cams, 9, generator=g) * 1e-2,
        "ray_o_noise": torch.randn(gh, gw, 3, generator=g),
        "ray_d_noise": torch.randn(gh, gw, 3, generator=g),
        "ray_o_noise_scale": 1e-3, "ray_d_noise_scale": 1e-3,
        "extrinsics_noise_scale": 1.0, "intrinsics_noise_scale": 1.0,
        "multiplicative_noise": multiplicative, "grid_size": grid_size,
    }


def keypoints(H: int, W: int, n: int, n_cams: int, seed=6, integer=False):
    """kps [n,2] float (x,y) uniform over the image (sub-pixel unless `integer`), with
    the corner pixels forced in; idx [n] int64 camera ids."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, generator=g) * (W - 1)
    y = torch.rand(n, generator=g) * (H - 1)
    kps = torch.stack([x, y], dim=-1)
    kps[0] = torch.tensor([0.0, 0.0])
    kps[1] = torch.tensor([W - 1.0, H - 1.0])
    kps[2] = torch.tensor([W - 1.0, 0.0])
    kps[3] = torch.tensor([0.0, H - 1.0])
    if integer:
        kps = kps.floor()
    idx = torch.randint(0, n_cams, (n,), generator=g)
    return kps, idx


def matched_keypoints(H: int, W: int, K, E0, E1, n: int, seed=8, noise_px=0.7, outlier_frac=0.25):
    """n matched key-point pairs for the cameras E0 / E1 (camera-to-world, NeRF axes: x right, y up,
    camera looks down -z; pixel = (fx X/-Z + cx, -fy Y/-Z + cy) as get_rays_kps_* implies): 3-D points
    in front of both cameras projected into each view + Gaussian pixel noise; `outlier_frac` of the
    pairs get an unrelated second key point (large error / negative depth cases)."""
    g = torch.Generator().manual_seed(seed)
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])

    def project(E, X):
        Xc = (X - E[:3, 3]) @ E[:3, :3]
        return torch.stack([fx * Xc[:, 0] / -Xc[:, 2] + cx, -fy * Xc[:, 1] / -Xc[:, 2] + cy], -1), -Xc[:, 2]
    pts, k0, k1 = [], [], []
    kept = 0
    while kept < n:
        X = torch.cat([torch.randn(4 * n, 2, generator=g) * 1.2, -3.0 - 2.0 * torch.rand(4 * n, 1, generator=g)], -1)
        a, za = project(E0, X)
        b, zb = project(E1, X)
        ok = (za > 0.5) & (zb > 0.5) & (a[:, 0] > 0) & (a[:, 0] < W - 1) & (a[:, 1] > 0) & (a[:, 1] < H - 1) \
            & (b[:, 0] > 0) & (b[:, 0] < W - 1) & (b[:, 1] > 0) & (b[:, 1] < H - 1)
        k0.append(a[ok]); k1.append(b[ok])
        kept += int(ok.sum())
    k0 = torch.cat(k0)[:n] + torch.randn(n, 2, generator=g) * noise_px
    k1 = torch.cat(k1)[:n] + torch.randn(n, 2, generator=g) * noise_px
    n_out = int(n * outlier_frac)
    k1[:n_out, 0] = torch.rand(n_out, generator=g) * (W - 1)
    k1[:n_out, 1] = torch.rand(n_out, generator=g) * (H - 1)
    k0[:, 0].clamp_(0, W - 1); k1[:, 0].clamp_(0, W - 1)
    k0[:, 1].clamp_(0, H - 1); k1[:, 1].clamp_(0, H - 1)
    return k0.contiguous(), k1.contiguous()


def nerfpp_params(seed: int = 777, n_freqs: int = 10, n_freqs_views: int = 4, D: int = 8, W: int = 256):
    """State dict of a freshly constructed NeRF++ `NerfNet` (fg_net then bg_net, each: D trunk layers,
    density layer, remap layer, two colour layers -- nerfplusplus/nerf_network.py:86-115) with torch's
    default nn.Linear initialisation drawn in the reference's construction order under
    torch.manual_seed(seed) -- pinned bit-for-bit by tests/golden/nerfpp.npz 'init/*'."""
    import torch.nn as nn
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    out = {}
    try:
        for prefix, in_dim in (("fg_net.", 3), ("bg_net.", 4)):
            in_ch = in_dim + 2 * in_dim * n_freqs
            in_views = 3 + 6 * n_freqs_views
            dim = in_ch
            layers = []
            for i in range(D):
                layers.append(("base_layers.%d.0" % i, nn.Linear(dim, W)))
                dim = W
                if i == 4 and i != D - 1:
                    dim += in_ch
            layers.append(("sigma_layers.0", nn.Linear(dim, 1)))
            layers.append(("base_remap_layers.0", nn.Linear(dim, 256)))
            layers.append(("rgb_layers.0", nn.Linear(256 + in_views, W // 2)))
            layers.append(("rgb_layers.2", nn.Linear(W // 2, 3)))
            for name, lin in layers:
                out[prefix + name + ".weight"] = lin.weight.detach().clone()
                out[prefix + name + ".bias"] = lin.bias.detach().clone()
    finally:
        torch.random.set_rng_state(state)
    return out


def nerfpp_rays(n: int, seed: int = 21):
    """Rays of a camera inside the unit sphere (NeRF++ normalises scenes so): origins |o| <= 0.45,
    directions of length ~1-1.4 (z = 1 before rotation, as K^-1 p gives), min_depth 1e-4."""
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(n, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * (0.1 + 0.35 * torch.rand(n, 1, generator=g))
    d = torch.cat([torch.rand(n, 2, generator=g) * 1.2 - 0.6, torch.ones(n, 1)], -1)
    q = torch.randn(3, 3, generator=g)
    q, _ = torch.linalg.qr(q)
    d = d @ q.t()
    return o.contiguous(), d.contiguous(), torch.full((n,), 1e-4)


def nerfpp_randoms(n: int, s0: int, s1: int, seed: int = 22):
    """Injected uniforms of one two-level cascade step: level-0 jitter of the fg / bg depths, level-1
    inverse-CDF draws."""
    g = torch.Generator().manual_seed(seed)
    return {"t_fg": torch.rand(n, s0, generator=g), "t_bg": torch.rand(n, s0, generator=g),
            "u_fg": torch.rand(n, s1, generator=g), "u_bg": torch.rand(n, s1, generator=g)}


def camera_model(H: int, W: int, n_cams=17, seed=4, key="pinhole_rot_noise_10k_rayo_rayd", multiplicative=True,
                 grid_size=10, focal=400.0):
    """A learnable camera model (scnerf_amd.camera_model classes through camera_dict) over `camera_spec`'s rig,
    with the spec's non-zero residuals copied in so every gradient path is live.  -> (module on CPU, spec)."""
    import types
    from .camera_dict import camera_dict
    spec = camera_spec(H, W, n_cams=n_cams, seed=seed, multiplicative=multiplicative, grid_size=grid_size, focal=focal)
    args = types.SimpleNamespace(camera_model=key, grid_size=grid_size, ray_o_noise_scale=spec["ray_o_noise_scale"],
                                 ray_d_noise_scale=spec["ray_d_noise_scale"],
                                 extrinsics_noise_scale=spec["extrinsics_noise_scale"],
                                 intrinsics_noise_scale=spec["intrinsics_noise_scale"],
                                 multiplicative_noise=multiplicative, distortion_noise_scale=1e-2)
    cm = camera_dict[key](spec["K_init"], list(spec["poses"].numpy()), args, H, W)
    with torch.no_grad():
        cm.intrinsics_noise.copy_(spec["intrinsics_noise"])
        cm.extrinsics_noise.copy_(spec["extrinsics_noise"])
        cm.ray_o_noise.copy_(spec["ray_o_noise"])
        if cm.ray_o_noise.data_ptr() != cm.ray_d_noise.data_ptr():
            cm.ray_d_noise.copy_(spec["ray_d_noise"])
    return cm, spec


# ---- a procedural scene with analytic density and colour (PSNR trajectories: tools/psnr_trajectory.py) -------------
def procedural_field(x: Tensor, d: Tensor):
    """density [..] and colour [.., 3] of a smooth synthetic scene at points x [.., 3] seen along unit directions d:
    three Gaussian blobs of different colour inside the unit ball, a mild view-dependent tint."""
    centres = torch.tensor([[0.35, 0.1, 0.0], [-0.3, -0.25, 0.2], [0.0, 0.35, -0.3]], dtype=x.dtype)
    widths = torch.tensor([0.28, 0.22, 0.25], dtype=x.dtype)
    peaks = torch.tensor([18.0, 25.0, 14.0], dtype=x.dtype)
    base = torch.tensor([[0.9, 0.25, 0.2], [0.2, 0.8, 0.3], [0.25, 0.35, 0.9]], dtype=x.dtype)
    r2 = ((x[..., None, :] - centres) ** 2).sum(-1)                       # [.., 3]
    w = peaks * torch.exp(-0.5 * r2 / widths ** 2)
    sigma = w.sum(-1)
    colour = (w[..., None] * base).sum(-2) / (sigma[..., None] + 1e-9)
    tint = 0.1 * torch.sin(3.0 * x + 2.0 * d)                             # view dependence
    return sigma, (colour + tint).clamp(0.02, 0.98)


def procedural_rays(n_views=24, res=32, seed=11, near=2.5, far=5.5) -> Tensor:
    """[n_views * res * res, 11] ray batch (o, d, near, far, unit view direction): pinhole cameras on a sphere of radius
    4 looking at the origin."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for v in range(n_views):
        c = torch.randn(3, generator=g, dtype=torch.float64)
        c = 4.0 * c / c.norm()
        fwd = -c / c.norm()
        up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
        right = torch.linalg.cross(fwd, up)
        right = right / right.norm()
        up = torch.linalg.cross(right, fwd)
        ii, jj = torch.meshgrid(torch.linspace(-0.35, 0.35, res, dtype=torch.float64),
                                torch.linspace(-0.35, 0.35, res, dtype=torch.float64), indexing="ij")
        d = fwd + ii[..., None] * right + jj[..., None] * up
        d = d.reshape(-1, 3)
        o = c.expand_as(d)
        vd = d / d.norm(dim=-1, keepdim=True)
        nf = torch.tensor([near, far], dtype=torch.float64).expand(d.shape[0], 2)
        out.append(torch.cat([o, d, nf, vd], -1))
    return torch.cat(out).float()


def procedural_targets(rays: Tensor, n_quad=768) -> Tensor:
    """the scene's pixel colours: the volume-rendering integral by a dense fp64 quadrature along every ray"""
    r = rays.double()
    o, d, near, far, vd = r[:, 0:3], r[:, 3:6], r[:, 6:7], r[:, 7:8], r[:, 8:11]
    t = torch.linspace(0.0, 1.0, n_quad, dtype=torch.float64)
    z = near * (1 - t) + far * t
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    sigma, colour = procedural_field(pts, vd[:, None, :].expand_as(pts))
    delta = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1) * d.norm(dim=-1, keepdim=True)
    alpha = 1.0 - torch.exp(-sigma * delta)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    return ((alpha * trans)[..., None] * colour).sum(-2).float()
