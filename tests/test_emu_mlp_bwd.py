"""Fused dgrad chain (HIP source under the CPU SIMT interpreter) vs torch autograd on the
oracle network."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import scnerf_oracle as O
from scnerf_amd import mlp_layout as ML
from scnerf_amd import synthetic as synth
from tests.emu import harness as H
from tests.emu_mlp_util import network_params, pack_forward, pack_backward, grad_views

pytestmark = pytest.mark.emu


def oracle_backward(p, pts, vd, spr, d_raw):
    """autograd on the oracle formulas with every pre-activation retained."""
    pts = pts.clone().requires_grad_(True)
    vd = vd.clone().requires_grad_(True)
    P = pts.shape[0]
    n_rays = P // spr
    vps = vd[:, None, :].expand(n_rays, spr, 3).reshape(P, 3)
    e = O.positional_encoding(pts, 10)
    ev = O.positional_encoding(vps, 4)
    zs = []
    h = e
    for i in range(8):
        z = F.linear(h, p["pts_linears.%d.weight" % i], p["pts_linears.%d.bias" % i])
        z.retain_grad()
        zs.append(z)
        h = F.relu(z)
        if i == 4:
            h = torch.cat([e, h], -1)
    sigma = F.linear(h, p["alpha_linear.weight"], p["alpha_linear.bias"])
    feat = F.linear(h, p["feature_linear.weight"], p["feature_linear.bias"])
    feat.retain_grad()
    zv = F.linear(torch.cat([feat, ev], -1), p["views_linears.0.weight"], p["views_linears.0.bias"])
    zv.retain_grad()
    rgb = F.linear(F.relu(zv), p["rgb_linear.weight"], p["rgb_linear.bias"])
    raw = torch.cat([rgb, sigma], -1)
    (raw * d_raw).sum().backward()
    return dict(dz=[z.grad for z in zs], dfeat=feat.grad, dzv=zv.grad, d_pts=pts.grad, d_vd=vd.grad)


@pytest.mark.parametrize("pd", [3, 4])
@pytest.mark.parametrize("n_rays,spr", [(5, 32), (1, 70)])
def test_mlp_dgrad_matches_autograd(n_rays, spr, pd):
    lay = ML.layout(pd)
    p = network_params(2 if pd == 3 else 778, pd)
    wpk, wbk = pack_forward(p, pd), pack_backward(p, pd)
    P = n_rays * spr
    g = torch.Generator().manual_seed(9)
    pts = torch.rand(P, pd, generator=g) * 2.4 - 1.2
    vd = torch.randn(n_rays, 3, generator=g)
    vd = vd / vd.norm(dim=-1, keepdim=True)
    d_raw = torch.randn(P, 4, generator=g)
    raw = np.zeros((P, 4), np.float32)
    save = np.full(lay.save_floats(P), np.nan, np.float32)
    H.call("scnerf_mlp_fwd", pd, pts.numpy(), vd.numpy(), 3, spr, wpk, raw, save, P, None)
    grads = np.full(ML.grad_floats(P), np.nan, np.float32)
    d_pts = np.full((P, pd), np.nan, np.float32)
    d_views = np.full((P, 3), np.nan, np.float32)
    H.call("scnerf_mlp_bwd", pd, d_raw.numpy(), pts.numpy(), vd.numpy(), 3, spr, wbk, save, grads, d_pts, d_views, P, None)
    ref = oracle_backward(p, pts, vd, spr, d_raw)
    gv = grad_views(grads, P)

    def sec(name, w):
        return gv[name]

    def close(a, b, what):
        scale = float(np.abs(b).max()) + 1e-12
        err = float(np.abs(a - b).max())
        assert err <= 2e-5 * scale + 1e-7, "%s: max err %g vs scale %g" % (what, err, scale)

    close(sec("dzv", 128), ref["dzv"].numpy(), "dzv")
    close(sec("dfeat", 256), ref["dfeat"].numpy(), "dfeat")
    for l in range(7, -1, -1):
        close(sec("dz%d" % l, 256), ref["dz"][l].numpy(), "dz%d" % l)
    close(d_pts, ref["d_pts"].numpy(), "d_pts")
    close(d_views.reshape(n_rays, spr, 3).sum(1), ref["d_vd"].numpy(), "d_viewdirs")


@pytest.mark.parametrize("pd", [3, 4])
def test_full_network_weight_gradients_match_autograd(pd):
    """fwd (train) -> dgrad -> scnerf_nerf_wgrad: every parameter gradient of the network (no chunk maxima here: every
    GEMM on the exact-fp32 MFMA; the three-fp16-product GEMMs are tests/test_emu_mlp_h3.py's)."""
    _full_network_weight_gradients(pd)


def _full_network_weight_gradients(pd):
    lay = ML.layout(pd)
    p = {k: v.clone().requires_grad_(True) for k, v in network_params(4 if pd == 3 else 779, pd).items()}
    pdet = {k: v.detach() for k, v in p.items()}
    wpk, wbk = pack_forward(pdet, pd), pack_backward(pdet, pd)
    n_rays, spr = 3, 50
    P = n_rays * spr
    g = torch.Generator().manual_seed(12)
    pts = torch.rand(P, pd, generator=g) * 2.4 - 1.2
    vd = torch.randn(n_rays, 3, generator=g)
    vd = vd / vd.norm(dim=-1, keepdim=True)
    d_raw = torch.randn(P, 4, generator=g)
    raw = np.zeros((P, 4), np.float32)
    save = np.full(lay.save_floats(P), np.nan, np.float32)
    H.call("scnerf_mlp_fwd", pd, pts.numpy(), vd.numpy(), 3, spr, wpk, raw, save, P, None)
    grads = np.full(ML.grad_floats(P), np.nan, np.float32)
    d_pts = np.zeros((P, pd), np.float32)
    d_views = np.zeros((P, 3), np.float32)
    H.call("scnerf_mlp_bwd", pd, d_raw.numpy(), pts.numpy(), vd.numpy(), 3, spr, wbk, save, grads, d_pts, d_views, P, None)
    chunks = 3
    ws = np.full(H.lib().scnerf_nerf_wgrad_workspace_floats(chunks), np.nan, np.float32)
    flat = np.full(lay.n_params, np.nan, np.float32)
    assert H.lib().scnerf_nerf_param_count(pd) == lay.n_params
    H.call("scnerf_nerf_wgrad", pd, save, grads, d_raw.numpy(), P, chunks, ws, flat, 0, None)
    twice = flat.copy()
    H.call("scnerf_nerf_wgrad", pd, save, grads, d_raw.numpy(), P, chunks, ws, twice, 1, None)     # accumulate
    np.testing.assert_allclose(twice, 2.0 * flat, rtol=1e-6, atol=1e-30)
    out = O.query_network(p, pts.reshape(n_rays, spr, pd), vd).reshape(P, 4)
    (out * d_raw).sum().backward()
    assert not np.isnan(flat).any()
    for name, shape in lay.param_shapes:
        o = lay.param_offsets[name]
        got = flat[o:o + int(np.prod(shape))].reshape(shape)
        ref = p[name].grad.numpy()
        scale = float(np.abs(ref).max()) + 1e-12
        err = float(np.abs(got - ref).max())
        assert err <= 3e-5 * scale + 1e-6, "%s: err %g scale %g" % (name, err, scale)
