"""Pins oracle/nerfpp_oracle.py (the CPU restatement of the NeRF++ path) against
golden vectors produced by the reference's own nerfplusplus/ code (oracle/gen_golden.py:gen_nerfpp)."""
import os

import numpy as np
import pytest
import torch

from oracle import nerfpp_oracle as NO
from oracle import scnerf_oracle as O
from scnerf_amd import synthetic as synth

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "nerfpp.npz"))
GS = np.load(os.path.join(os.path.dirname(__file__), "golden", "nerfpp_sampler.npz"))     # oracle/gen_golden.py:gen_nerfpp_sampler


def T(k):
    return torch.from_numpy(G[k])


def close(a, b, rtol, what, atol=0.0):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    scale = float(np.abs(b).max()) + 1e-30
    err = float(np.abs(a - b).max())
    assert err <= rtol * scale + atol, "%s: err %g scale %g" % (what, err, scale)


def check_fingerprints(prefix, named_grads, rtol=2e-3):
    """norm + random projection of every parameter gradient (fp32 graphs of ~1e5 terms: 2e-3)."""
    for name, g in named_grads:
        ref_norm, ref_dot = G[prefix + name]
        norm, dot = NO.grad_fingerprint(name, g)
        assert abs(norm - ref_norm) <= rtol * ref_norm + 1e-12, (name, norm, ref_norm)
        assert abs(dot - ref_dot) <= rtol * ref_norm * 3 + 1e-12, (name, dot, ref_dot)      # |dot| ~ norm


def test_init_matches_reference_construction():
    p = synth.nerfpp_params(777)
    sums = np.array([float(v.double().abs().sum()) for v in p.values()])
    np.testing.assert_array_equal(sums, G["init/abs_sums"])
    np.testing.assert_array_equal(p["fg_net.base_layers.0.0.weight"].reshape(-1)[:8].numpy(), G["init/first8_fg0"])
    np.testing.assert_array_equal(p["bg_net.base_layers.5.0.weight"].reshape(-1)[:8].numpy(), G["init/first8_bg5"])
    assert p["bg_net.base_layers.0.0.weight"].shape == (256, 84) and p["bg_net.base_layers.5.0.weight"].shape == (256, 340)


def test_function_kats():
    o, d = T("kat/ray_o"), T("kat/ray_d")
    close(NO.intersect_sphere(o, d), G["kat/far"], 1e-6, "far")
    np.testing.assert_allclose(NO.perturb_samples(T("kat/z"), T("kat/t_rand")).numpy(), G["kat/perturbed"], rtol=1e-6, atol=1e-7)
    s = NO.sample_pdf(T("kat/bins"), T("kat/weights"), T("kat/u"))
    np.testing.assert_allclose(s.numpy(), G["kat/pdf_samples"], rtol=2e-6, atol=2e-6)
    u_det = torch.linspace(0.0, 1.0, 40).expand(48, 40)
    np.testing.assert_allclose(NO.sample_pdf(T("kat/bins"), T("kat/weights"), u_det).numpy(), G["kat/pdf_det"],
                               rtol=2e-6, atol=2e-6)
    oo, dd = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    depth = T("kat/bg_depth")
    pts, dreal = NO.depth2pts_outside(oo[:, None].expand(48, 16, 3), dd[:, None].expand(48, 16, 3), depth)
    np.testing.assert_allclose(pts.detach().numpy(), G["kat/bg_pts"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(dreal.detach().numpy(), G["kat/bg_depth_real"], rtol=2e-5, atol=1e-5)
    (pts * T("kat/bg_g_pts")).sum().backward()
    close(oo.grad, G["kat/bg_g_o"], 2e-4, "g_o")
    close(dd.grad, G["kat/bg_g_d"], 2e-4, "g_d")


def test_nerfnet_forward_and_gradients():
    p = {k: v.clone().requires_grad_(True) for k, v in synth.nerfpp_params(778).items()}
    o, d = T("fwd/ray_o").requires_grad_(True), T("fwd/ray_d").requires_grad_(True)
    n = o.shape[0]
    near = torch.full((n,), 1e-4)
    far = NO.intersect_sphere(o, d)
    close(far, G["fwd/far"], 1e-6, "far")
    fg_z = near[:, None] + T("fwd/frac") * (far - near)[:, None]
    ret = NO.nerfnet_forward(p, o, d, far, fg_z, T("fwd/bg_z"))
    for name in ("rgb", "fg_weights", "bg_weights", "fg_rgb", "fg_depth", "bg_rgb", "bg_depth", "bg_lambda"):
        np.testing.assert_allclose(ret[name].detach().numpy(), G["fwd/ret/" + name], rtol=2e-4, atol=2e-6, err_msg=name)
    loss = ((ret["rgb"] - T("fwd/target")) ** 2).mean() + (ret["fg_weights"] * T("fwd/gw")).sum() \
        + ret["bg_depth"].mean() * 0.1 + ret["fg_depth"].mean() * 0.1
    assert abs(float(loss.detach()) - float(G["fwd/loss"])) <= 1e-5 * abs(float(G["fwd/loss"]))
    loss.backward()
    close(o.grad, G["fwd/g_ray_o"], 1e-3, "g_ray_o")
    close(d.grad, G["fwd/g_ray_d"], 1e-3, "g_ray_d")
    check_fingerprints("fwd/gproj/", [(k, v.grad) for k, v in p.items()])
    for name in ("fg_net.base_layers.0.0.weight", "bg_net.base_layers.0.0.weight", "bg_net.sigma_layers.0.weight",
                 "fg_net.rgb_layers.2.weight", "fg_net.rgb_layers.2.bias", "bg_net.base_remap_layers.0.bias"):
        close(p[name].grad, G["fwd/g/" + name], 1e-3, name)


def test_two_level_cascade_step():
    n, s0, s1 = 32, 64, 128
    rnd = synth.nerfpp_randoms(n, s0, s1, seed=26)
    ps = [{k: v.clone().requires_grad_(True) for k, v in synth.nerfpp_params(sd).items()} for sd in (779, 780)]
    o, d = T("step/ray_o").requires_grad_(True), T("step/ray_d").requires_grad_(True)
    near = torch.full((n,), 1e-4)
    far, fg0, bg0 = NO.cascade_depths_level0(o, d, near, s0, rnd["t_fg"], rnd["t_bg"])
    np.testing.assert_allclose(fg0.detach().numpy(), G["step/fg_depth0"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(bg0.detach().numpy(), G["step/bg_depth0"], rtol=2e-6, atol=1e-7)
    ret0 = NO.nerfnet_forward(ps[0], o, d, far, fg0, bg0)
    fg1, bg1 = NO.cascade_depths_next(fg0, bg0, ret0["fg_weights"], ret0["bg_weights"], rnd["u_fg"], rnd["u_bg"])
    # samples move when a weight-dependent index flips; the bulk must agree tightly
    for got, key in ((fg1, "step/fg_depth1"), (bg1, "step/bg_depth1")):
        e = np.abs(got.detach().numpy() - G[key])
        assert (e < 1e-5).mean() > 0.995, (key, (e < 1e-5).mean())
    ret1 = NO.nerfnet_forward(ps[1], o, d, far, fg1, bg1)
    target = T("step/target")
    loss = ((ret0["rgb"] - target) ** 2).mean() + ((ret1["rgb"] - target) ** 2).mean()
    np.testing.assert_allclose(ret0["rgb"].detach().numpy(), G["step/rgb0"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(ret1["rgb"].detach().numpy(), G["step/rgb1"], rtol=0, atol=1e-4)
    assert abs(float(loss.detach()) - float(G["step/loss"])) <= 1e-4 * float(G["step/loss"])
    loss.backward()
    close(o.grad, G["step/g_ray_o"], 5e-3, "g_ray_o")
    close(d.grad, G["step/g_ray_d"], 5e-3, "g_ray_d")
    for lvl in (0, 1):
        check_fingerprints("step/gproj%d/" % lvl, [(k, v.grad) for k, v in ps[lvl].items()], rtol=5e-3)


@pytest.mark.parametrize("tag", ["plain", "dist"])
def test_pixel_centre_ray_generator(tag):
    from test_gpu_camera import _oracle_cam
    H, W = 60, 80
    spec = synth.camera_spec(H, W, n_cams=4, seed=33, multiplicative=True, focal=70.0)
    cam = _oracle_cam(spec, grad=True)
    k = "rays_%s/" % tag
    if tag == "dist":                       # the distortion class builds both noise grids from one tensor (A15 quirk)
        cam["ray_d_noise"] = cam["ray_o_noise"]
    fx, fy, cx, cy = O.camera_intrinsic_params(cam)
    K = torch.zeros(3, 3)
    K = torch.stack([torch.stack([fx, torch.zeros(()), cx]), torch.stack([torch.zeros(()), fy, cy]),
                     torch.tensor([0.0, 0.0, 1.0])])
    R, t = O.camera_extrinsics(cam)
    c2w = torch.cat([torch.cat([R[2], t[2][:, None]], 1), torch.tensor([[0.0, 0, 0, 1]])], 0)
    dist = None
    dist_noise = None
    if tag == "dist":
        dist_noise = torch.tensor([0.3, -0.2], requires_grad=True)
        dist = torch.from_numpy(G[k + "k"]) + dist_noise * 1e-1
    sel = torch.from_numpy(G[k + "select"])
    on = O.upsample_noise_grid(cam["ray_o_noise"], H, W, cam["ray_o_noise_scale"])
    dn = O.upsample_noise_grid(cam["ray_d_noise"], H, W, cam["ray_d_noise_scale"])
    ro, rd, dep = NO.camera_rays_pixel_centre(K, c2w, sel, H, W, dist, on, dn)
    np.testing.assert_allclose(ro.detach().numpy(), G[k + "rays_o"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rd.detach().numpy(), G[k + "rays_d"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dep.detach().numpy(), G[k + "depth"], rtol=1e-6)
    ((ro * T(k + "g_o")).sum() + (rd * T(k + "g_d")).sum()).backward()
    close(cam["intrinsics_noise"].grad, G[k + "g_intrinsics_noise"], 1e-3, "intrinsics_noise")
    close(cam["extrinsics_noise"].grad, G[k + "g_extrinsics_noise"], 1e-3, "extrinsics_noise")
    if tag == "plain":
        close(cam["ray_o_noise"].grad, G[k + "g_ray_o_noise"], 1e-4, "ray_o_noise")
        close(cam["ray_d_noise"].grad, G[k + "g_ray_d_noise"], 1e-3, "ray_d_noise")
    else:
        close(dist_noise.grad, G[k + "g_distortion_noise"], 1e-3, "distortion_noise")
