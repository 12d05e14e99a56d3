"""Camera ray generator kernels (HIP source under the CPU SIMT interpreter) vs golden vectors of
the reference's get_rays_* / CameraModel / ndc_rays* (values and autograd gradients)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import scnerf_oracle as O
from scnerf_amd import synthetic as synth
from tests.emu import harness as H

pytestmark = pytest.mark.emu
HH, WW = 378, 504


def cam_arrays(spec, aliased):
    poses = spec["poses"]
    K = spec["K_init"]
    a = dict(
        intr_init=np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]], np.float32),
        intr_noise=spec["intrinsics_noise"].numpy().astype(np.float32),
        extr_init=torch.cat([O.rotation_to_ortho6d(poses[:, :3, :3]), poses[:, :3, 3]], -1).numpy().astype(np.float32).copy(),
        extr_noise=spec["extrinsics_noise"].numpy().astype(np.float32),
        grid_o=spec["ray_o_noise"].numpy().astype(np.float32),
        grid_d=(spec["ray_o_noise"] if aliased else spec["ray_d_noise"]).numpy().astype(np.float32))
    return a


def close(a, b, tol, what):
    scale = float(np.abs(b).max()) + 1e-30
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, "%s: err %g scale %g" % (what, err, scale)


def fwd(a, spec, kps, idx, ext, n, single=0):
    ro = np.zeros((n, 3), np.float32); rd = np.zeros((n, 3), np.float32)
    gh, gw = a["grid_o"].shape[:2]
    H.call("scnerf_camera_rays_fwd", kps, idx, single, ext, 0 if ext is None else ext.shape[0] // 1 if ext.ndim == 3 else 1,
           a["intr_init"], a["intr_noise"], ctypes.c_float(spec["intrinsics_noise_scale"]), int(spec["multiplicative_noise"]),
           a["extr_init"], a["extr_noise"], ctypes.c_float(spec["extrinsics_noise_scale"]), a["extr_init"].shape[0],
           a["grid_o"], ctypes.c_float(spec["ray_o_noise_scale"]), a["grid_d"], ctypes.c_float(spec["ray_d_noise_scale"]),
           gh, gw, HH, WW, ro, rd, n, None)
    return ro, rd


def bwd(a, spec, kps, idx, ext, n, g_o, g_d, single=0):
    gh, gw = a["grid_o"].shape[:2]
    C = a["extr_init"].shape[0]
    n_ext = 0 if ext is None else (ext.shape[0] if ext.ndim == 3 else 1)
    out = dict(di=np.full(4, np.nan, np.float32), de=np.full((C, 9), np.nan, np.float32),
               dgo=np.full((gh, gw, 3), np.nan, np.float32), dgd=np.full((gh, gw, 3), np.nan, np.float32),
               dE=np.full((max(n_ext, 1), 4, 4), np.nan, np.float32))
    ws = np.zeros(H.lib().scnerf_camera_bwd_workspace_floats(max(C, n_ext)), np.float32)
    H.call("scnerf_camera_rays_bwd", kps, idx, single, ext, n_ext,
           a["intr_init"], a["intr_noise"], ctypes.c_float(spec["intrinsics_noise_scale"]), int(spec["multiplicative_noise"]),
           a["extr_init"], a["extr_noise"], ctypes.c_float(spec["extrinsics_noise_scale"]), C,
           a["grid_o"], ctypes.c_float(spec["ray_o_noise_scale"]), a["grid_d"], ctypes.c_float(spec["ray_d_noise_scale"]),
           gh, gw, HH, WW, g_o, g_d, out["di"], out["de"], out["dgo"], out["dgd"], out["dE"] if n_ext else None, ws, n, None)
    return out


@pytest.mark.parametrize("tag,mult,aliased", [("plain_add", False, False), ("plain_mul", True, False),
                                              ("dist_mul", True, True)])
def test_camera_rays_per_ray_cameras(golden, tag, mult, aliased):
    g = golden("camera")
    k = tag + "/"
    spec = synth.camera_spec(HH, WW, n_cams=5, seed=4, multiplicative=mult)
    a = cam_arrays(spec, aliased)
    kps, idx = g[k + "kps"], g[k + "idx"].astype(np.int64)
    n = kps.shape[0]
    ro, rd = fwd(a, spec, kps, idx, None, n)
    np.testing.assert_allclose(ro, g[k + "rays_o"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rd, g[k + "rays_d"], rtol=1e-5, atol=1e-6)
    o = bwd(a, spec, kps, idx, None, n, g[k + "g_o"], g[k + "g_d"])
    close(o["di"], g[k + "g_intrinsics_noise"], 1e-3, "d intrinsics_noise")
    close(o["de"], g[k + "g_extrinsics_noise"], 1e-3, "d extrinsics_noise")
    close(o["dgo"], g[k + "g_ray_o_noise"], 1e-4, "d ray_o_noise")
    close(o["dgd"], g[k + "g_ray_d_noise"], 1e-3, "d ray_d_noise")


def test_camera_rays_shared_extrinsic_and_ndc(golden):
    g = golden("camera")
    k = "plain_mul/"
    spec = synth.camera_spec(HH, WW, n_cams=5, seed=4, multiplicative=True)
    a = cam_arrays(spec, False)
    kps = g[k + "kps"]
    n = kps.shape[0]
    E = np.ascontiguousarray(g[k + "E"][2])                  # the camera model's own pose 2, as a plain matrix
    ro, rd = fwd(a, spec, kps, None, E, n)
    np.testing.assert_allclose(ro, g[k + "shared/rays_o"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rd, g[k + "shared/rays_d"], rtol=1e-5, atol=1e-6)
    # the same through the learnable pose of camera 2 (single index)
    ro2, rd2 = fwd(a, spec, kps, None, None, n, single=2)
    np.testing.assert_allclose(ro2, ro, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rd2, rd, rtol=1e-5, atol=1e-6)
    # NDC through the model's focal lengths + chained gradients
    K = g[k + "K"]
    f2 = np.array([K[0, 0], K[1, 1]], np.float32)
    no = np.zeros((n, 3), np.float32); nd = np.zeros((n, 3), np.float32)
    H.call("scnerf_ndc_fwd", HH, WW, f2, ctypes.c_float(1.0), ro2, rd2, no, nd, n, None)
    np.testing.assert_allclose(no, g[k + "shared/ndc_o"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(nd, g[k + "shared/ndc_d"], rtol=2e-5, atol=2e-6)
    g_ro = np.zeros((n, 3), np.float32); g_rd = np.zeros((n, 3), np.float32); g_f = np.zeros(2, np.float32)
    H.call("scnerf_ndc_bwd", HH, WW, f2, ctypes.c_float(1.0), ro2, rd2, g[k + "g_o"], g[k + "g_d"], g_ro, g_rd, g_f, n, None)
    o = bwd(a, spec, kps, None, None, n, g_ro, g_rd, single=2)
    # intrinsics get gradient through the rays AND through fx, fy of the warp (multiplicative residual)
    scale = spec["intrinsics_noise_scale"] * a["intr_init"]
    di = o["di"].copy()
    di[0] += g_f[0] * scale[0]
    di[1] += g_f[1] * scale[1]
    close(di, g[k + "shared/g_intrinsics_noise"], 2e-3, "d intrinsics_noise (rays + ndc)")
    close(o["dgo"], g[k + "shared/g_ray_o_noise"], 1e-3, "d ray_o_noise")
    # in the reference this branch received a detached pose tensor?  no: E came from get_extrinsic() -> gradient flows
    ref_de = g[k + "shared/g_extrinsics_noise"]
    close(o["de"], ref_de, 2e-3, "d extrinsics_noise")


def test_explicit_extrinsic_gradient_is_consistent(golden):
    """d/dE from the explicit-matrix branch == the accumulators the learnable branch feeds into
    Gram-Schmidt (finite-difference spot check on one entry)."""
    g = golden("camera")
    k = "plain_add/"
    spec = synth.camera_spec(HH, WW, n_cams=5, seed=4, multiplicative=False)
    a = cam_arrays(spec, False)
    kps = g[k + "kps"]
    n = kps.shape[0]
    E = np.ascontiguousarray(g[k + "E"][1]).astype(np.float32)
    go, gd = g[k + "g_o"], g[k + "g_d"]
    o = bwd(a, spec, kps, None, E, n, go, gd)

    def loss(Em):
        ro, rd = fwd(a, spec, kps, None, np.ascontiguousarray(Em.astype(np.float32)), n)
        return float((ro.astype(np.float64) * go).sum() + (rd.astype(np.float64) * gd).sum())
    for (r, c) in ((0, 1), (2, 3), (1, 2)):
        Ep, Em = E.astype(np.float64).copy(), E.astype(np.float64).copy()
        Ep[r, c] += 1e-2
        Em[r, c] -= 1e-2
        fd = (loss(Ep) - loss(Em)) / 2e-2
        assert abs(fd - o["dE"][0, r, c]) <= 2e-2 * max(1.0, abs(fd)), (r, c, fd, o["dE"][0, r, c])


def test_pinhole_and_full_image(golden):
    g = golden("camera")
    kps = g["pinhole/kps"]
    n = kps.shape[0]
    ro = np.zeros((n, 3), np.float32); rd = np.zeros((n, 3), np.float32)
    H.call("scnerf_pinhole_rays", kps, 2, g["pinhole/c2w"], ctypes.c_float(400.0), HH, WW, ro, rd, n, None)
    np.testing.assert_allclose(ro, g["pinhole/rays_o"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rd, g["pinhole/rays_d"], rtol=1e-6, atol=1e-7)
    f2 = np.array([400.0, 400.0], np.float32)
    no = np.zeros((n, 3), np.float32); nd = np.zeros((n, 3), np.float32)
    H.call("scnerf_ndc_fwd", HH, WW, f2, ctypes.c_float(1.0), ro, rd, no, nd, n, None)
    np.testing.assert_allclose(no, g["pinhole/ndc_o"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(nd, g["pinhole/ndc_d"], rtol=2e-5, atol=2e-6)
    # full image (kps = NULL) == key points enumerating every pixel
    h, w = 6, 9
    ro_f = np.zeros((h * w, 3), np.float32); rd_f = np.zeros((h * w, 3), np.float32)
    H.call("scnerf_pinhole_rays", None, 0, g["pinhole/c2w"], ctypes.c_float(11.0), h, w, ro_f, rd_f, h * w, None)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    kk = np.stack([xx.reshape(-1), yy.reshape(-1)], -1).astype(np.float32)
    ro_k = np.zeros_like(ro_f); rd_k = np.zeros_like(rd_f)
    H.call("scnerf_pinhole_rays", kk, 2, g["pinhole/c2w"], ctypes.c_float(11.0), h, w, ro_k, rd_k, h * w, None)
    np.testing.assert_array_equal(rd_f, rd_k)


def test_upsample_grid_matches_interpolate():
    g = torch.Generator().manual_seed(3)
    grid = torch.randn(37, 50, 3, generator=g)
    out = np.zeros((HH * WW, 3), np.float32)
    H.call("scnerf_upsample_grid_fwd", grid.numpy(), ctypes.c_float(1e-3), 37, 50, HH, WW, out, None)
    gt = grid.clone().requires_grad_(True)
    ref = torch.nn.functional.interpolate(gt.permute(2, 0, 1)[None], (HH, WW), mode="bilinear",
                                          align_corners=False).permute(0, 2, 3, 1).reshape(-1, 3) * 1e-3
    np.testing.assert_allclose(out, ref.detach().numpy(), rtol=1e-5, atol=1e-9)
    go = torch.randn(HH * WW, 3, generator=g)
    (ref * go).sum().backward()
    dg = np.zeros((37, 50, 3), np.float32)
    H.call("scnerf_upsample_grid_bwd", go.numpy(), ctypes.c_float(1e-3), 37, 50, HH, WW, dg, None)
    np.testing.assert_allclose(dg, gt.grad.numpy(), rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("ndc,use_viewdirs", [(True, True), (False, True), (True, False)])
def test_pack_ray_batch_matches_the_reference_composition(ndc, use_viewdirs):
    """render()'s view-direction / NDC / concatenation steps (reference render.py:105-128) as the one fused
    launch, against the same steps written with torch ops on the oracle's NDC warp, with gradients to the rays
    and to the two focal lengths."""
    from tests.emu.host_on_emu import emulated_device
    from oracle import scnerf_oracle as O
    from scnerf_amd import camera_functional as CF
    H, W, n = 48, 64, 37
    g = torch.Generator().manual_seed(3)
    o = torch.randn(n, 3, generator=g) * 0.3
    d = torch.cat([torch.randn(n, 2, generator=g) * 0.4, -1.0 - torch.rand(n, 1, generator=g)], -1)
    f2 = torch.tensor([55.0, 57.0])
    gy = torch.randn(n, 11 if use_viewdirs else 8, generator=g)

    class Cam:                                   # the only thing pack_ray_batch asks of a camera model
        def __init__(self, f):
            self.f = f

        def focal_xy(self):
            return self.f
    with emulated_device():
        oe, de, fe = o.clone().requires_grad_(True), d.clone().requires_grad_(True), f2.clone().requires_grad_(True)
        got = CF.pack_ray_batch(H, W, oe, de, 0.25, 3.0, use_viewdirs, ndc, camera_model=Cam(fe))
        (got * gy).sum().backward()
    orf, drf, frf = o.clone().requires_grad_(True), d.clone().requires_grad_(True), f2.clone().requires_grad_(True)
    cols = []
    if use_viewdirs:
        cols.append(drf / torch.norm(drf, dim=-1, keepdim=True))
    ro, rd = O.ndc_rays(H, W, frf[0], frf[1], 1.0, orf, drf) if ndc else (orf, drf)
    ones = torch.ones_like(rd[:, :1])
    want = torch.cat([ro, rd, 0.25 * ones, 3.0 * ones] + cols, -1)
    (want * gy).sum().backward()
    np.testing.assert_allclose(got.detach().numpy(), want.detach().numpy(), rtol=2e-6, atol=2e-6)
    for a, b, what in ((oe, orf, "rays_o"), (de, drf, "rays_d")):
        assert float((a.grad - b.grad).abs().max()) <= 2e-5 * float(b.grad.abs().max()) + 1e-7, what
    if ndc:
        assert float((fe.grad - frf.grad).abs().max()) <= 2e-5 * float(frf.grad.abs().max())
    else:
        assert fe.grad is None


@pytest.mark.parametrize("mult", [False, True])
def test_camera_matrices_match_the_tensor_ops(mult):
    """get_intrinsic() / get_extrinsic() as one launch each way (scnerf_camera_matrices_fwd / _bwd) against the tensor
    ops the camera model runs on CPU parameters (camera_utils.ortho2rotation, intrinsic_param_to_K and torch's autograd
    through them): values to 1e-6, gradients for random upstream gradients to 1e-5 of the largest entry, incl. a camera
    whose second axis is (nearly) parallel to the first."""
    from scnerf_amd.camera_utils import get_44_rotation_matrix_from_33_rotation_matrix, intrinsic_param_to_K, ortho2rotation
    g = torch.Generator().manual_seed(4)
    C = 7
    intr_init = torch.tensor([400.0, 410.0, 250.0, 190.0])
    intr_noise = (torch.randn(4, generator=g) * 3.0).requires_grad_(True)
    extr_init = torch.randn(C, 9, generator=g)
    extr_init[3, 3:6] = extr_init[3, 0:3] * 1.5 + 1e-3 * torch.randn(3, generator=g)
    extr_noise = (torch.randn(C, 9, generator=g) * 0.05).requires_grad_(True)
    si, se = 0.7, 0.3
    p = intr_init + intr_noise * si * (intr_init if mult else 1.0)
    K_ref = intrinsic_param_to_K(p)
    E_ref = get_44_rotation_matrix_from_33_rotation_matrix(ortho2rotation(extr_init[:, :6] + se * extr_noise[:, :6]))
    E_ref[..., :3, 3] = extr_init[:, 6:] + se * extr_noise[:, 6:]
    gK, gE = torch.randn(4, 4, generator=g), torch.randn(C, 4, 4, generator=g)
    (K_ref * gK).sum().backward(retain_graph=True)
    (E_ref * gE).sum().backward()
    K = np.full((4, 4), np.nan, np.float32)
    E = np.full((C, 4, 4), np.nan, np.float32)
    f32 = lambda t_: t_.detach().numpy().astype(np.float32).copy()
    H.call("scnerf_camera_matrices_fwd", f32(intr_init), f32(intr_noise), ctypes.c_float(si), int(mult), f32(extr_init),
           f32(extr_noise), ctypes.c_float(se), C, K, E, None)
    close(K, f32(K_ref), 1e-6, "K")
    close(E, f32(E_ref), 1e-6, "E")
    d_in = np.full(4, np.nan, np.float32)
    d_ex = np.full((C, 9), np.nan, np.float32)
    H.call("scnerf_camera_matrices_bwd", f32(intr_init), ctypes.c_float(si), int(mult), f32(extr_init), f32(extr_noise),
           ctypes.c_float(se), C, f32(gK), f32(gE), d_in, d_ex, None)
    close(d_in, f32(intr_noise.grad), 1e-5, "d intrinsics_noise")
    close(d_ex, f32(extr_noise.grad), 1e-5, "d extrinsics_noise")
    # absent upstream gradients are zeros
    d_ex2 = np.full((C, 9), np.nan, np.float32)
    H.call("scnerf_camera_matrices_bwd", f32(intr_init), ctypes.c_float(si), int(mult), f32(extr_init), f32(extr_noise),
           ctypes.c_float(se), C, f32(gK), None, d_in, d_ex2, None)
    assert not d_ex2.any()


def test_intrinsic_and_extrinsic_share_one_node_until_its_backward():
    """CameraModel.get_intrinsic() / get_extrinsic() (model/camera_model.py:160-192) of the same parameter values come from
    ONE CameraMatricesFunction node (one launch each way for the pair); the pair is dropped when the node's backward has
    run or a parameter changed; and an in-place parameter update between forward and backward is caught by autograd's
    version check (save_for_backward) instead of silently differentiating at the new values."""
    import types
    from tests.emu.host_on_emu import emulated_device
    from scnerf_amd.camera_dict import camera_dict
    H, W = 48, 64
    spec = synth.camera_spec(H, W, n_cams=3, seed=4, multiplicative=True)
    args = types.SimpleNamespace(camera_model="pinhole_rot_noise_10k_rayo_rayd", grid_size=10,
                                 ray_o_noise_scale=spec["ray_o_noise_scale"], ray_d_noise_scale=spec["ray_d_noise_scale"],
                                 extrinsics_noise_scale=spec["extrinsics_noise_scale"],
                                 intrinsics_noise_scale=spec["intrinsics_noise_scale"], multiplicative_noise=True)
    with emulated_device():
        cm = camera_dict["pinhole_rot_noise_10k_rayo_rayd"](spec["K_init"], list(spec["poses"].numpy()), args, H, W)
        with torch.no_grad():
            cm.intrinsics_noise.copy_(spec["intrinsics_noise"])
            cm.extrinsics_noise.copy_(spec["extrinsics_noise"])
        # the default is the reference's structure: a graph per getter call
        assert cm.get_intrinsic().grad_fn is not cm.get_extrinsic().grad_fn
        cm.share_matrix_node = True                                       # (what dropin.install() and bench.py opt into)
        K, E = cm.get_intrinsic(), cm.get_extrinsic()
        assert K.grad_fn is E.grad_fn and K.grad_fn is not None
        assert cm.get_intrinsic() is K and cm.get_extrinsic() is E
        with torch.no_grad():
            assert cm.get_intrinsic() is not K                          # another grad mode: another entry
        cam = O.camera_state(spec, grad=True)
        Rw, tw = O.camera_extrinsics(cam)
        fx, fy, cx, cy = O.camera_intrinsic_params(cam)
        gK = torch.randn(4, 4, generator=torch.Generator().manual_seed(1))
        gE = torch.randn(3, 4, 4, generator=torch.Generator().manual_seed(2))
        K, E = cm.get_intrinsic(), cm.get_extrinsic()
        ((K * gK).sum() + (E * gE).sum()).backward()
        ((fx * gK[0, 0] + fy * gK[1, 1] + cx * gK[0, 2] + cy * gK[1, 2]) + (Rw * gE[:, :3, :3]).sum()
         + (tw * gE[:, :3, 3]).sum()).backward()
        close(cm.intrinsics_noise.grad.numpy(), cam["intrinsics_noise"].grad.numpy(), 1e-5, "d intrinsics_noise")
        close(cm.extrinsics_noise.grad.numpy(), cam["extrinsics_noise"].grad.numpy(), 1e-4, "d extrinsics_noise")
        K2 = cm.get_intrinsic()                                         # the graph behind K is gone: a fresh node
        assert K2 is not K and torch.equal(K2, K)
        (K2.sum() + cm.get_extrinsic().sum()).backward()                # ... that can be differentiated again
        K3 = cm.get_intrinsic()
        with torch.no_grad():
            cm.intrinsics_noise.add_(0.01)
        K4 = cm.get_intrinsic()
        assert K4 is not K3 and not torch.equal(K4, K3)                 # a new parameter version: recomputed
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            K3.sum().backward()
