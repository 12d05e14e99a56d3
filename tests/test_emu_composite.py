"""Compositing kernels (HIP source under the CPU SIMT interpreter) vs the reference's golden
vectors (values and autograd gradients of raw2outputs)."""
import numpy as np
import pytest
import torch

from tests.emu import harness as H

pytestmark = pytest.mark.emu


def rel_close(a, b, tol, what):
    scale = float(np.abs(b).max()) + 1e-30
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, "%s: max err %g (scale %g)" % (what, err, scale)


@pytest.mark.parametrize("tag", ["s64", "s192"])
@pytest.mark.parametrize("wb", [0, 1])
@pytest.mark.parametrize("with_noise", [0, 1])
def test_composite_forward_backward_golden(golden, tag, wb, with_noise):
    g = golden("composite")
    raw, z, rays_d = g[tag + "/raw"], g[tag + "/z"], g[tag + "/rays_d"]
    n, s = z.shape
    rays = np.zeros((n, 8), np.float32)
    rays[:, 3:6] = rays_d
    noise = g[tag + "/noise"] if with_noise else None
    rgb = np.zeros((n, 3), np.float32); disp = np.zeros(n, np.float32); acc = np.zeros(n, np.float32)
    depth = np.zeros(n, np.float32); w = np.zeros((n, s), np.float32)
    H.call("scnerf_composite_fwd", raw, z, rays, 8, noise, wb, rgb, disp, acc, depth, w, n, s, None)
    key = "%s/wb%d_n%d/" % (tag, wb, with_noise)
    # alpha = 1 - exp(-a*dist) cancels for tiny a*dist: a 1-ulp difference of expf (libm vs ATen) is 6e-8 absolute
    np.testing.assert_allclose(w, g[key + "weights"], rtol=2e-6, atol=1.5e-7)
    np.testing.assert_allclose(rgb, g[key + "rgb"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(acc, g[key + "acc"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(depth, g[key + "depth"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(disp, g[key + "disp"], rtol=2e-5)
    d_raw = np.full((n, s, 4), np.nan, np.float32)
    d_rd = np.full((n, 3), np.nan, np.float32)
    H.call("scnerf_composite_bwd", raw, z, rays, 8, noise, wb, g[tag + "/g_rgb"], g[tag + "/g_disp"],
           g[tag + "/g_acc"], g[tag + "/g_depth"], None, d_raw, d_rd, n, s, None)
    ref = g[key + "g_raw"]
    for r in range(n):                       # per ray: gradients span many orders of magnitude
        rel_close(d_raw[r], ref[r], 2e-4, "d_raw ray %d" % r)
    rel_close(d_rd, g[key + "g_rays_d"], 2e-4, "d_rays_d")


def test_composite_backward_extra_raw_gradient_and_null_inputs(golden):
    g = golden("composite")
    raw, z, rays_d = g["s64/raw"], g["s64/z"], g["s64/rays_d"]
    n, s = z.shape
    rays = np.zeros((n, 11), np.float32)
    rays[:, 3:6] = rays_d
    a = np.zeros((n, s, 4), np.float32); b = np.zeros((n, s, 4), np.float32)
    extra = np.random.default_rng(0).standard_normal((n, s, 4)).astype(np.float32)
    H.call("scnerf_composite_bwd", raw, z, rays, 11, None, 0, g["s64/g_rgb"], None, None, None, None, a, None, n, s, None)
    H.call("scnerf_composite_bwd", raw, z, rays, 11, None, 0, g["s64/g_rgb"], None, None, None, extra, b, None, n, s, None)
    np.testing.assert_allclose(b, a + extra, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("s", [64, 192, 70])
def test_ray_reduce(s):
    rng = np.random.default_rng(s)
    n = 9
    d_pts = rng.standard_normal((n, s, 3)).astype(np.float32)
    d_views = rng.standard_normal((n, s, 3)).astype(np.float32)
    z = rng.random((n, s)).astype(np.float32)
    extra = rng.standard_normal((n, 3)).astype(np.float32)
    out = np.full((n, 11), np.nan, np.float32)
    H.call("scnerf_ray_reduce", d_pts, d_views, z, extra, out, 11, 0, n, s, None)
    exp = np.zeros((n, 11))
    exp[:, 0:3] = d_pts.astype(np.float64).sum(1)
    exp[:, 3:6] = (d_pts.astype(np.float64) * z[..., None]).sum(1) + extra
    exp[:, 8:11] = d_views.astype(np.float64).sum(1)
    np.testing.assert_allclose(out, exp, rtol=1e-5, atol=1e-5)
    H.call("scnerf_ray_reduce", d_pts, d_views, z, None, out, 11, 1, n, s, None)
    exp2 = exp.copy()
    exp2[:, 0:6] += exp[:, 0:6]
    exp2[:, 3:6] -= extra
    exp2[:, 8:11] += exp[:, 8:11]
    np.testing.assert_allclose(out, exp2, rtol=1e-5, atol=1e-5)
