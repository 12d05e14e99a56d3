"""GPU parity tests (run on the MI355X box: pytest -m gpu): the HIP kernels, called through
the C ABI, against the CPU oracle on the same seeded inputs and against the committed golden
vectors of the reference."""
import numpy as np
import pytest
import torch

from oracle import scnerf_oracle as O
from scnerf_amd import mlp_layout as ML
from scnerf_amd import synthetic as synth
from conftest import t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from scnerf_amd import ops as _ops
    _ops.check_layout()
    return _ops


def dev(x):
    return x.contiguous().cuda()


@pytest.mark.parametrize("tag", ["rand", "knot", "det"])
def test_sample_pdf_golden_bit_exact(ops, golden, tag):
    g = golden("sample_pdf")
    u = t(g[tag + "/u"])
    if tag == "det":
        u = u[0].contiguous()
    s, inds, cdf = ops.sample_pdf(dev(t(g["bins"])), dev(t(g["weights"])), dev(u), True, True)
    np.testing.assert_array_equal(cdf.cpu().numpy(), g[tag + "/cdf"])
    np.testing.assert_array_equal(inds.cpu().numpy(), g[tag + "/inds"])
    np.testing.assert_array_equal(s.cpu().numpy(), g[tag + "/samples"])
    assert inds.dtype == torch.int64


@pytest.mark.parametrize("side", ["left", "right"])
@pytest.mark.parametrize("Ba,Bv,A,V", [(1, 1, 1, 1), (1, 100, 50, 12), (100, 1, 50, 12), (200, 200, 1, 120),
                                       (100, 100, 63, 128), (100, 100, 500, 120)])
def test_searchsorted_matches_numpy(ops, side, Ba, Bv, A, V):
    rng = np.random.default_rng(Ba + Bv + A + V)
    a = np.sort(rng.random((Ba, A), dtype=np.float32), -1)
    v = rng.random((Bv, V), dtype=np.float32)
    out = ops.searchsorted(dev(t(a)), dev(t(v)), side).cpu().numpy()
    for r in range(max(Ba, Bv)):
        np.testing.assert_array_equal(out[r], np.searchsorted(a[0 if Ba == 1 else r], v[0 if Bv == 1 else r], side=side))


@pytest.mark.parametrize("lindisp,perturb", [(0, 0), (0, 1), (1, 1)])
def test_coarse_sample_bit_exact(ops, lindisp, perturb):
    n, s = 1001, 64
    rays = synth.ray_batch(n, seed=5, lindisp=bool(lindisp))
    t_rand = synth.render_randoms(n, s, 0, seed=6)["t_rand"]
    t_vals = torch.linspace(0.0, 1.0, steps=s)
    z, pts = ops.coarse_sample(dev(rays), dev(t_vals), dev(t_rand) if perturb else None, bool(lindisp))
    zo = O.stratified_z(rays[:, 6:7], rays[:, 7:8], s, bool(lindisp), t_rand if perturb else None)
    po = rays[:, None, 0:3] + rays[:, None, 3:6] * zo[:, :, None]
    np.testing.assert_array_equal(z.cpu().numpy(), zo.numpy())
    np.testing.assert_array_equal(pts.cpu().numpy(), po.numpy())


@pytest.mark.parametrize("sc,sf,det,n", [(64, 128, False, 1023), (64, 64, False, 100), (64, 128, True, 100)])
def test_fine_sample_bit_exact(ops, sc, sf, det, n):
    g = torch.Generator().manual_seed(sc + sf)
    rays = synth.ray_batch(n, seed=7)
    z_c = O.stratified_z(rays[:, 6:7], rays[:, 7:8], sc, False, torch.rand(n, sc, generator=g))
    w_c = torch.rand(n, sc, generator=g) ** 5
    w_c[1] = 0.0
    w_c[2, : sc // 2] = 0.0
    u = O.deterministic_u(n, sf) if det else torch.rand(n, sf, generator=g)
    ud = dev(u[0]) if det else dev(u)
    z_f, pts_f, z_s, z_std, inds, cdf = ops.fine_sample(dev(rays), dev(z_c), dev(w_c), ud, True, True)
    z_mid = 0.5 * (z_c[:, 1:] + z_c[:, :-1])
    so, io, co = O.sample_pdf(z_mid, w_c[:, 1:-1], u.contiguous(), rowsum="aten")
    zf_o = torch.sort(torch.cat([z_c, so], -1), -1)[0]
    pts_o = rays[:, None, 0:3] + rays[:, None, 3:6] * zf_o[:, :, None]
    np.testing.assert_array_equal(cdf.cpu().numpy(), co.numpy())
    np.testing.assert_array_equal(inds.cpu().numpy(), io.numpy())
    np.testing.assert_array_equal(z_s.cpu().numpy(), so.numpy())
    np.testing.assert_array_equal(z_f.cpu().numpy(), zf_o.numpy())
    np.testing.assert_array_equal(pts_f.cpu().numpy(), pts_o.numpy())
    np.testing.assert_allclose(z_std.cpu().numpy(), torch.std(so, -1, unbiased=False).numpy(), rtol=1e-5, atol=1e-8)


def _flat(p, pd=3):
    return torch.cat([p[name].reshape(-1) for name, _ in ML.layout(pd).param_shapes])


@pytest.mark.parametrize("resident", [False, True], ids=["fused_fp32", "resident"])
@pytest.mark.parametrize("pd", [3, 4])
@pytest.mark.parametrize("n_rays,spr,save", [(37, 64, True), (16, 192, False)])
def test_mlp_forward_matches_oracle(ops, n_rays, spr, save, pd, resident):
    """pd = 3: the SCNeRF network; pd = 4: NeRF++'s background network (points x, y, z, 1/r); the fused fp32 kernel and
    the resident-arithmetic kernel (three fp16 products, csrc/mlp_fwd_h3.hip), training and inference instantiations."""
    from tests.emu_mlp_util import network_params, oracle_activations
    lay = ML.layout(pd)
    p = network_params(0 if pd == 3 else 777, pd)
    wpk = ops.pack_weights(dev(_flat(p, pd)), "fwd", pd=pd)
    idx = lay.forward_index()
    src = _flat(p, pd).numpy()
    np.testing.assert_array_equal(wpk.cpu().numpy(), np.where(idx >= 0, src[np.maximum(idx, 0)], 0).astype(np.float32))
    P = n_rays * spr
    g = torch.Generator().manual_seed(3)
    pts = torch.rand(P, pd, generator=g) * 3 - 1.5
    vd = torch.randn(n_rays, 3, generator=g)
    vd = vd / vd.norm(dim=-1, keepdim=True)
    sv = torch.full((lay.save_floats(P),), float("nan"), device="cuda") if save else None
    planes = ops.pack_resident(dev(_flat(p, pd)), pd) if resident else None
    raw = ops.mlp_fwd(dev(pts), dev(vd), spr, wpk, sv, pd=pd, planes=planes).cpu()
    ref = O.query_network(p, pts.reshape(n_rays, spr, pd), vd).reshape(P, 4)
    np.testing.assert_allclose(raw.numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
    if save:
        from tests.emu_mlp_util import save_views
        vps = vd[:, None, :].expand(n_rays, spr, 3).reshape(P, 3)
        oa = oracle_activations(p, pts, vps)
        s = save_views(sv.cpu().numpy(), P, pd)
        np.testing.assert_allclose(s["epts"][:, :lay.in_pts], oa["e"].numpy(), rtol=0, atol=2e-6)
        assert np.all(s["epts"][:, lay.in_pts:] == 0)
        np.testing.assert_allclose(s["eviews"][:, :27], oa["ev"].numpy(), rtol=0, atol=2e-6)
        for l in range(8):
            np.testing.assert_allclose(s["act%d" % l], oa["acts"][l].numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(s["feat"], oa["feat"].numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(s["hv"], oa["hv"].numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("resident", [False, True], ids=["fused_fp32", "resident"])
@pytest.mark.parametrize("pd,kind", [(3, "xavier"), (4, "xavier"), (3, "trained")])
def test_mlp_backward_and_weight_gradients_match_autograd(ops, pd, resident, kind):
    """(`kind`: tests/trained_weights.py)  train forward -> dgrad -> 12 wgrad GEMMs for both network variants vs torch autograd on the oracle; forward and
    data gradients on the fused fp32 kernels or on the resident-arithmetic ones."""
    from tests.emu_mlp_util import network_params
    lay = ML.layout(pd)
    from tests import trained_weights as TW
    src = network_params(4 if pd == 3 else 779, pd) if kind == "xavier" else TW.weights(kind, 4)
    p = {k: v.clone().requires_grad_(True) for k, v in src.items()}
    flat = dev(_flat({k: v.detach() for k, v in p.items()}, pd))
    n_rays, spr = 21, 50                       # 1050 samples: ragged last workgroup
    P = n_rays * spr
    g = torch.Generator().manual_seed(12)
    pts = (torch.rand(P, pd, generator=g) * 2.4 - 1.2).requires_grad_(True)
    vd = torch.randn(n_rays, 3, generator=g)
    vd = (vd / vd.norm(dim=-1, keepdim=True)).requires_grad_(True)
    d_raw = torch.randn(P, 4, generator=g)
    save = ops.save_workspace(P, "cuda", pd)
    planes = ops.pack_resident(flat, pd) if resident else None
    raw = ops.mlp_fwd(dev(pts.detach()), dev(vd.detach()), spr, ops.pack_weights(flat, "fwd", pd=pd), save, pd=pd, planes=planes)
    grads, d_pts, d_views = ops.mlp_bwd(dev(d_raw), dev(pts.detach()), dev(vd.detach()), spr,
                                        ops.pack_weights(flat, "bwd", pd=pd), save, pd=pd, planes=planes)
    fg = ops.nerf_wgrad(save, grads, dev(d_raw), P, pd=pd).cpu().numpy()
    out = O.query_network(p, pts.reshape(n_rays, spr, pd), vd).reshape(P, 4)
    np.testing.assert_allclose(raw.cpu().numpy(), out.detach().numpy(), rtol=2e-5, atol=2e-5)
    (out * d_raw).sum().backward()

    def close(a, b, tol, what):
        scale = float(np.abs(b).max()) + 1e-12
        err = float(np.abs(a - b).max())
        assert err <= tol * scale + 1e-6, "%s: err %g scale %g" % (what, err, scale)
    # A pre-activation within an ulp of zero can land on the other side of the ReLU than on the CPU; that
    # sample's point gradient (amplified 2^9 by the encoding) then differs.  Per-sample: nearly all rows
    # tight; sums over samples (view / weight gradients): loose.
    ref_pts = pts.grad.numpy()
    row_err = np.abs(d_pts.cpu().numpy() - ref_pts).max(1)
    assert (row_err <= 1e-4 * np.abs(ref_pts).max() + 1e-6).mean() >= 0.99, (row_err > 1e-4 * np.abs(ref_pts).max()).sum()
    def grad_close(a, b, what, q=0.999, tol_q=1e-3, tol_max=5e-2):
        scale = float(np.abs(b).max()) + 1e-12
        err = np.abs(a - b).reshape(-1)
        assert np.quantile(err, q) <= tol_q * scale + 1e-7 and err.max() <= tol_max * scale + 1e-6, \
            "%s: q%.3f err %g, max err %g, scale %g" % (what, q, np.quantile(err, q), err.max(), scale)
    grad_close(d_views.cpu().numpy().reshape(n_rays, spr, 3).sum(1), vd.grad.numpy(), "d_views", q=0.9, tol_q=2e-3)
    assert np.isfinite(fg).all()
    for name, shape in lay.param_shapes:
        o = lay.param_offsets[name]
        grad_close(fg[o:o + int(np.prod(shape))].reshape(shape), p[name].grad.numpy(), name, q=0.99, tol_q=2e-3)


def _gates_from_masks(mask_sec, P, ntile):
    """lane-native ReLU bit words [tiles, 64 lanes, 4 words] -> bool [P, 32 ntile] (element 16 t + r of lane (m, h) =
    feature feat_of(t, r, h): word i >> 5, bit 31 - (i & 31))"""
    m = np.asarray(mask_sec, dtype=np.uint32)
    out = np.zeros((m.shape[0] * 32, 32 * ntile), bool)
    for t_ in range(ntile):
        for r in range(16):
            i = 16 * t_ + r
            bit = (m[:, :, i >> 5] >> np.uint32(31 - (i & 31))) & np.uint32(1)           # [tiles, 64]
            for hh in range(2):
                out[:, ML.feat_of(t_, r, hh)] = bit[:, 32 * hh: 32 * hh + 32].reshape(-1).astype(bool)
    return out[:P]


@pytest.mark.parametrize("resident", [False, True], ids=["fused_fp32", "resident"])
def test_relu_gate_flips_are_attributed(ops, resident):
    """Gradient differences against autograd on the CPU, attributed like the sampler's (tests/parity_attribution.py):
    the only discontinuity of the network is the ReLU gate, and a pre-activation within rounding of zero may land on
    either side in two fp32 evaluations.  (1) every (sample, unit) whose gate differs between the kernel's saved bit
    masks and the CPU run has a CPU pre-activation within a few roundings of zero -- relative to sum |w x| + |b|;
    (2) with the KERNEL's gates imposed on the CPU run, everything agrees tightly: raw to 2e-5, every row of d pts to
    2e-5 of the largest entry, every parameter gradient to 2e-5 (q0.999) / 1e-4 (max) of its largest entry (measured:
    1e-6 and 5e-6, profiles/parity_r03.json) -- a hundred to five hundred times tighter than the unattributed bounds of
    test_mlp_backward_and_weight_gradients_match_autograd, which have to absorb the flipped samples."""
    import torch.nn.functional as F
    from tests import parity_attribution as PA
    from tests.emu_mlp_util import network_params, save_views
    pd = 3
    lay = ML.layout(pd)
    p = {k: v.clone().requires_grad_(True) for k, v in network_params(4, pd).items()}
    flat = dev(_flat({k: v.detach() for k, v in p.items()}, pd))
    n_rays, spr = 64, 192
    P = n_rays * spr
    g = torch.Generator().manual_seed(21)
    pts = (torch.rand(P, pd, generator=g) * 2.4 - 1.2).requires_grad_(True)
    vd = torch.randn(n_rays, 3, generator=g)
    vd = (vd / vd.norm(dim=-1, keepdim=True))
    d_raw = torch.randn(P, 4, generator=g)
    save = ops.save_workspace(P, "cuda", pd)
    planes = ops.pack_resident(flat, pd) if resident else None
    raw = ops.mlp_fwd(dev(pts.detach()), dev(vd), spr, ops.pack_weights(flat, "fwd", pd=pd), save, pd=pd, planes=planes)
    grads, d_pts, d_views = ops.mlp_bwd(dev(d_raw), dev(pts.detach()), dev(vd), spr, ops.pack_weights(flat, "bwd", pd=pd), save,
                                        pd=pd, planes=planes)
    fg = ops.nerf_wgrad(save, grads, dev(d_raw), P, pd=pd).cpu().numpy()
    sv = save_views(save.cpu().numpy(), P, pd)
    gates = [torch.from_numpy(_gates_from_masks(sv["mask"][l], P, 8)) for l in range(8)]
    gate_v = torch.from_numpy(_gates_from_masks(sv["mask"][8], P, 4))

    vps = vd[:, None, :].expand(n_rays, spr, 3).reshape(P, 3)

    def forward(impose):
        e = O.positional_encoding(pts, 10)
        ev = O.positional_encoding(vps, 4)
        h, zs = e, []
        for i in range(8):
            z = F.linear(h, p["pts_linears.%d.weight" % i], p["pts_linears.%d.bias" % i])
            zs.append((z, h))
            h = z * gates[i] if impose else F.relu(z)
            if i == 4:
                h = torch.cat([e, h], -1)
        sigma = F.linear(h, p["alpha_linear.weight"], p["alpha_linear.bias"])
        feat = F.linear(h, p["feature_linear.weight"], p["feature_linear.bias"])
        xin = torch.cat([feat, ev], -1)
        zv = F.linear(xin, p["views_linears.0.weight"], p["views_linears.0.bias"])
        zs.append((zv, xin))
        hv = zv * gate_v if impose else F.relu(zv)
        rgb = F.linear(hv, p["rgb_linear.weight"], p["rgb_linear.bias"])
        return torch.cat([rgb, sigma], -1), zs

    # (1) where the gates differ, the CPU pre-activation is within rounding of zero
    with torch.no_grad():
        _, zs = forward(False)
    names = ["pts_linears.%d" % i for i in range(8)] + ["views_linears.0"]
    n_flips, worst = 0, 0.0
    for (z, x), gate, name in zip(zs, gates + [gate_v], names):
        flip = (z > 0) != gate
        n_flips += int(flip.sum())
        if flip.any():
            size = x.abs() @ p[name + ".weight"].detach().abs().T + p[name + ".bias"].detach().abs()
            worst = max(worst, float((z.abs() / size)[flip].max()))
    assert worst <= 4e-6, worst                   # a few fp32 roundings of the accumulated sum
    # (2) the kernel's gates imposed on the CPU run
    out, _ = forward(True)
    np.testing.assert_allclose(raw.cpu().numpy(), out.detach().numpy(), rtol=2e-5, atol=2e-5)
    (out * d_raw).sum().backward()
    ref_pts = pts.grad.numpy()
    row_err = np.abs(d_pts.cpu().numpy() - ref_pts).max(1) / np.abs(ref_pts).max()
    assert row_err.max() <= 2e-5, row_err.max()
    report = {"gate_flips": n_flips, "gates": int(sum(g_.numel() for g_ in gates) + gate_v.numel()),
              "largest_flipped_preactivation_over_sum_abs": worst, "d_pts_worst_row": float(row_err.max()), "parameters": {}}
    for name, shape in lay.param_shapes:
        o = lay.param_offsets[name]
        a, b = fg[o:o + int(np.prod(shape))].reshape(-1), p[name].grad.numpy().reshape(-1)
        scale = float(np.abs(b).max()) + 1e-30
        err = np.abs(a - b) / scale
        report["parameters"][name] = {"q0.999": float(np.quantile(err, 0.999)), "max": float(err.max())}
        assert np.quantile(err, 0.999) <= 2e-5 and err.max() <= 1e-4, (name, report["parameters"][name])
    PA.REPORT["relu_gate_attribution/" + ("resident" if resident else "fused_fp32")] = report


def test_half_weight_gradient_gemm_is_fp32_grade(ops):
    """The 256 x 256 weight-gradient GEMM on three fp16 products (operands scaled per workgroup chunk and cut into two fp16
    numbers; csrc/wgrad256_half.h) against fp64, beside the exact-fp32 MFMA kernel on the same operands: full 24-bit
    significands, magnitudes spread over 2^8, half of X zero (post-ReLU).  Its error must be the fp32 kernel's
    (accumulation-order noise), not a reduced-precision one (a plain fp16 product would be at 5e-4 of sum |a b|)."""
    from scnerf_amd import _capi
    from tests import parity_attribution as PA
    lib = _capi.load()
    P, chunks = 65536, 256
    g = torch.Generator(device="cuda").manual_seed(3)
    def operand(relu):
        v = torch.randn(P, 256, device="cuda", generator=g) * torch.exp2(torch.randint(-4, 4, (P, 256), device="cuda", generator=g).float())
        if relu:
            v = v * (torch.rand(P, 256, device="cuda", generator=g) < 0.5)
        return v.contiguous()
    A, B = operand(False), operand(True)
    tiled = lambda m: m.reshape(P // 32, 32, 8, 4, 2, 4).permute(0, 2, 3, 4, 1, 5).contiguous().reshape(-1)
    At, Bt = tiled(A), tiled(B)
    ref = A.double().T @ B.double()
    scale = A.double().abs().T @ B.double().abs()
    ws = torch.empty(lib.scnerf_wgrad_workspace_floats(256, 256, chunks), device="cuda")
    err, out = {}, {}
    try:
        for mode in ("fp32",):
            assert ops.wgrad_arithmetic(mode) == mode
            dW = torch.full((256, 256), float("nan"), device="cuda")
            db = torch.full((256,), float("nan"), device="cuda")
            _capi.check(lib.scnerf_wgrad(ops._p(At), 256, 256, 256, 1, ops._p(Bt), 256, 256, 256, 1, P, chunks, ops._p(ws),
                                         ops._p(dW), 256, 0, ops._p(db), ops._stream()), "scnerf_wgrad")
            e = (dW.double() - ref).abs() / scale
            err[mode] = {"max": float(e.max()), "rms": float((e * e).mean().sqrt())}
            out[mode] = dW
            np.testing.assert_allclose(db.cpu().numpy(), A.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    finally:
        ops.wgrad_arithmetic("half")
    assert ops.wgrad_arithmetic() == "half"
    # three fp16 products with one scale per operand and workgroup chunk (csrc/wgrad256_half.h), maxima as the
    # resident kernels leave them; also with every sample's gradient scaled by its own power of two over 2^40
    chunk = lib.scnerf_wgrad_chunk_samples(P, chunks)
    ws2 = torch.empty(chunks * (65536 + 256), device="cuda")
    for tag, Az in (("half", A), ("half_wide_range", A * torch.exp2(torch.randint(-30, 10, (P, 1), device="cuda", generator=g).float()))):
        ref_z = Az.double().T @ B.double()
        scale_z = Az.double().abs().T @ B.double().abs()
        amax_a = Az.abs().view(chunks, -1).max(1)[0].contiguous()
        amax_b = B.abs().view(chunks, -1).max(1)[0].contiguous()
        assert chunk * chunks == P
        dW = torch.full((256, 256), float("nan"), device="cuda")
        db = torch.full((256,), float("nan"), device="cuda")
        _capi.check(lib.scnerf_wgrad256_half(ops._p(tiled(Az)), ops._p(Bt), P, chunks, ops._p(ws2), ops._p(dW), ops._p(db),
                                             ops._p(amax_a), ops._p(amax_b), ops._stream()), "scnerf_wgrad256_half")
        e = (dW.double() - ref_z).abs() / scale_z
        err[tag] = {"max": float(e.max()), "rms": float((e * e).mean().sqrt())}
        eb = (db.double() - Az.double().sum(0)).abs() / Az.double().abs().sum(0)           # fp32 sums of 65 536 terms
        assert float(eb.max()) <= 2e-6, (tag, float(eb.max()))
    PA.REPORT["wgrad256_arithmetic_65536_samples"] = {"error_over_sum_abs_products_vs_fp64": err}
    assert err["half"]["max"] < 2e-7, err
    assert err["half"]["max"] <= 2.0 * err["fp32"]["max"] + 1e-9 and err["half"]["rms"] <= 2.0 * err["fp32"]["rms"], err
    # (a few samples carry each sum when magnitudes differ by 2^40: bounded by one cut product, 3 x 2^-22)
    assert err["half_wide_range"]["max"] <= 3 * 2.0 ** -22 and err["half_wide_range"]["rms"] <= 1e-7, err


@pytest.mark.parametrize("shape", ["256x64", "128x256"])
def test_narrow_weight_gradient_gemms_are_fp32_grade(ops, shape):
    """The narrow weight-gradient GEMMs on three fp16 products (csrc/wgrad_half_narrow.h; 256 x 64: dZ x encoded point,
    row-major X with 63 valid columns; 128 x 256: views-layer dZ x feature) against fp64, beside the fp32-MFMA kernel
    (csrc/wgrad_tiles.h) on the same operands: the error must be that kernel's, as for the 256 x 256 GEMMs.  Maxima per
    coarse chunk of 8 workgroups, as the resident kernels leave them."""
    from scnerf_amd import _capi
    from tests import parity_attribution as PA
    lib = _capi.load()
    wa, wb = (int(v) for v in shape.split("x"))
    x_tiled = wb == 256
    k_out = wb if x_tiled else wb - 1
    P, chunks, n_coarse = 65536, 256, 32
    g = torch.Generator(device="cuda").manual_seed(7)
    A = (torch.randn(P, wa, device="cuda", generator=g) * torch.exp2(torch.randint(-4, 4, (P, wa), device="cuda", generator=g).float())).contiguous()
    if x_tiled:
        B = torch.randn(P, wb, device="cuda", generator=g) * torch.exp2(torch.randint(-4, 4, (P, wb), device="cuda", generator=g).float())
    else:
        B = torch.sin(torch.randn(P, wb, device="cuda", generator=g) * 40.0)           # an encoding: values in [-1, 1]
    B = B.contiguous()
    tiled = lambda m: m.reshape(P // 32, 32, m.shape[1] // 32, 4, 2, 4).permute(0, 2, 3, 4, 1, 5).contiguous().reshape(-1)
    At, Bin = tiled(A), (tiled(B) if x_tiled else B.reshape(-1))
    ref = A.double().T @ B.double()[:, :k_out]
    scale = A.double().abs().T @ B.double().abs()[:, :k_out]
    ws = torch.empty(lib.scnerf_wgrad_workspace_floats(wa, wb, chunks), device="cuda")
    err = {}
    saved = ops.wgrad_arithmetic()
    try:
        ops.wgrad_arithmetic("fp32")
        dW = torch.full((wa, k_out), float("nan"), device="cuda")
        db = torch.full((wa,), float("nan"), device="cuda")
        _capi.check(lib.scnerf_wgrad(ops._p(At), wa, wa, wa, 1, ops._p(Bin), wb, wb, k_out, int(x_tiled), P, chunks, ops._p(ws),
                                     ops._p(dW), k_out, 0, ops._p(db), ops._stream()), "scnerf_wgrad")
        e = (dW.double() - ref).abs() / scale
        err["fp32"] = {"max": float(e.max()), "rms": float((e * e).mean().sqrt())}
    finally:
        ops.wgrad_arithmetic(saved)
    coarse = P // n_coarse
    amax_a = A.abs().view(n_coarse, -1).max(1)[0].contiguous()
    amax_b = B.abs().view(n_coarse, -1).max(1)[0].contiguous()
    dW = torch.full((wa, k_out), float("nan"), device="cuda")
    db = torch.full((wa,), float("nan"), device="cuda")
    _capi.check(lib.scnerf_wgrad_half_narrow(ops._p(At), wa, ops._p(Bin), wb, k_out, int(x_tiled), P, chunks, ops._p(ws),
                                             ops._p(dW), ops._p(db), ops._p(amax_a), ops._p(amax_b), n_coarse, coarse,
                                             ops._stream()), "scnerf_wgrad_half_narrow")
    e = (dW.double() - ref).abs() / scale
    err["half"] = {"max": float(e.max()), "rms": float((e * e).mean().sqrt())}
    eb = (db.double() - A.double().sum(0)).abs() / A.double().abs().sum(0)
    assert float(eb.max()) <= 2e-6, float(eb.max())
    PA.REPORT["wgrad_narrow_%s_arithmetic_65536_samples_error_over_sum_abs_products_vs_fp64" % shape] = err
    assert err["half"]["max"] <= 2.0 * err["fp32"]["max"] + 1e-9 and err["half"]["rms"] <= 2.0 * err["fp32"]["rms"], err


@pytest.mark.parametrize("kind", ["xavier", "trained", "adversarial"])
def test_resident_layers_are_fp32_grade(ops, kind):
    """(`kind`: the weight distribution, tests/trained_weights.py -- the reference's initialisation, a network trained for
    5000 steps, and hand-made layers aimed at the per-sample scale bound.)  Every layer of the resident-arithmetic forward (three fp16 products, per-sample scale from the row 1-norm bound;
    csrc/mlp_h3.h) over 131 072 samples against fp64 ON ITS OWN INPUT (what the kernel saved for the layer below),
    beside the fused fp32-MFMA kernel judged the same way: the error relative to sum |w x| + |b| must be that of the
    exact-fp32 path (accumulation-order noise), for layer 0 (encoded point), the trunk, the skip layer, the linear
    feature layer and the views layer alike.  Also: the scales never overflow fp16 (no inf / nan anywhere) on inputs
    whose magnitude varies over 2^20 between samples."""
    from tests import parity_attribution as PA
    from tests import trained_weights as TW
    lay = ML.layout(3)
    p = TW.weights(kind, 4)
    flat = dev(_flat(p, 3))
    n_rays, spr = 2048, 64
    P = n_rays * spr
    g = torch.Generator().manual_seed(8)
    pts = torch.rand(P, 3, generator=g) * 2.4 - 1.2
    pts[: P // 8] *= torch.exp2(torch.randint(-10, 10, (P // 8, 1), generator=g).float())      # far / tiny points
    pts_host = pts.clone()
    pts = dev(pts)
    vd = torch.randn(n_rays, 3, generator=g)
    vd = dev(vd / vd.norm(dim=-1, keepdim=True))
    wf = ops.pack_weights(flat, "fwd")
    saves = {"fp32": ops.save_workspace(P, "cuda").zero_(), "resident": ops.save_workspace(P, "cuda").zero_()}
    raw32 = ops.mlp_fwd(pts, vd, spr, wf, saves["fp32"])
    planes = ops.pack_resident(flat)
    raw16 = ops.mlp_fwd(pts, vd, spr, wf, saves["resident"], planes=planes)
    assert bool(torch.isfinite(raw16).all()) and bool(torch.isfinite(saves["resident"][: lay.save_floats_per_sample * ML.padded_samples(P)]).all())
    Pp = ML.padded_samples(P)
    off, _ = ML.section_offsets(lay.save_sections, P)

    def rows(save, name, width=256):
        blk = save[off[name]: off[name] + width * Pp]
        if name in ML.TILED_SECTIONS:
            return blk.view(Pp // 32, width // 32, 4, 2, 32, 4).permute(0, 4, 1, 2, 3, 5).reshape(Pp, width)[:P]
        return blk.view(Pp, width)[:P]

    def layer_io(save, l):
        e = rows(save, "epts", lay.e_width)[:, :lay.in_pts]
        if l == "feat":
            return rows(save, "act7"), dev(p["feature_linear.weight"]), dev(p["feature_linear.bias"]), rows(save, "feat"), False
        if l == "views":
            x = torch.cat([rows(save, "feat"), rows(save, "eviews", 32)[:, :27]], 1)
            return x, dev(p["views_linears.0.weight"]), dev(p["views_linears.0.bias"]), rows(save, "hv", 128), True
        x = e if l == 0 else (torch.cat([e, rows(save, "act4")], 1) if l == 5 else rows(save, "act%d" % (l - 1)))
        return x, dev(p["pts_linears.%d.weight" % l]), dev(p["pts_linears.%d.bias" % l]), rows(save, "act%d" % l), True

    report = {}
    for l in (0, 1, 2, 3, 4, 5, 7, "feat", "views"):
        for mode in ("fp32", "resident"):
            x, W, b, z, relu = layer_io(saves[mode], l)
            ref = x.double() @ W.double().T + b.double()
            if relu:
                ref = torch.relu(ref)
            scale = x.double().abs() @ W.double().abs().T + b.double().abs()
            e = (z.double() - ref).abs() / scale
            report.setdefault("layer_%s" % l, {})[mode] = {"max": float(e.max()), "rms": float((e * e).mean().sqrt())}
    e_raw = (raw16 - raw32).abs().max(1)[0] / raw32.abs().max(1)[0].clamp_min(1.0)
    report["raw_vs_fused_fp32_relative_max"] = float(e_raw.max())
    # where the per-sample scales put the values they cut: (largest |value| of a sample's layer output) x S, log2
    margins = TW.scale_margins(ops, planes, saves["resident"], P, pts_host.cuda())
    report["per_sample_scale_margins_log2_of_max_abs_z_times_S"] = margins
    suffix = "" if kind == "xavier" else "_%s_weights" % kind
    PA.REPORT["resident_layer_arithmetic_131072_samples_error_over_sum_abs_products_vs_fp64" + suffix] = report
    for layer, m in margins.items():
        # fp32 grade needs max|z| S in [2^-3, 2^13) (csrc/mlp_h3.h:23-25): never above (fp16 would overflow), and below only
        # for samples whose whole layer is (nearly) dead
        assert m["log2_max"] < 13.0, (layer, m)
    if kind == "adversarial":
        # the cancellation row did what it was built for: layer 3's scales sit ~ten octaves lower than layer 2's
        assert margins["layer_3"]["log2_max"] < margins["layer_2"]["log2_max"] - 6.0, margins
    for layer, r in report.items():
        if not isinstance(r, dict) or "resident" not in r:
            continue
        assert r["resident"]["rms"] <= 2.0 * r["fp32"]["rms"] and r["resident"]["max"] <= 3.0 * r["fp32"]["max"] + 1e-9, (layer, r)
        assert r["resident"]["max"] < 1e-6, (layer, r)
    if kind == "adversarial":
        # an ill-conditioned network by construction: two correct fp32 evaluations of it differ by more than 2e-5 at the
        # output -- judged against fp64 instead (the oracle in double precision on a slice of the samples): the resident
        # arithmetic may be no further from it than the fused fp32 kernels are
        n_sub = 256
        p64 = {k: v.double() for k, v in p.items()}
        ref64 = O.query_network(p64, pts_host[: n_sub * spr].double().reshape(n_sub, spr, 3), vd[:n_sub].cpu().double()).reshape(-1, 4)
        size = ref64.abs().max(1)[0].clamp_min(1.0)
        e32 = float(((raw32[: n_sub * spr].cpu().double() - ref64).abs().max(1)[0] / size).max())
        e16 = float(((raw16[: n_sub * spr].cpu().double() - ref64).abs().max(1)[0] / size).max())
        report["raw_vs_fp64_relative_max"] = {"fp32": e32, "resident": e16}
        PA.REPORT["resident_layer_arithmetic_131072_samples_error_over_sum_abs_products_vs_fp64" + suffix] = report
        assert e16 <= 3.0 * e32 + 1e-6, report["raw_vs_fp64_relative_max"]
    else:
        assert report["raw_vs_fused_fp32_relative_max"] <= 2e-5, report


@pytest.mark.parametrize("kind", ["xavier", "trained", "adversarial"])
def test_resident_data_gradients_follow_the_fused_chain_row_by_row(ops, kind):
    """(`kind`: tests/trained_weights.py; the hand-made layers of "adversarial" inflate the column-1-norm bounds of the
    transposed layers just as they inflate the forward's row bounds.)  The resident data-gradient chain against the fused fp32 chain on the SAME saved activations, with every
    sample's incoming gradient scaled by its own power of ten between 1e-30 and 1e+10 (and some exactly zero): every
    gradient section agrees row by row to 2e-5 of the row's size -- the per-sample scales carry forty orders of
    magnitude -- and zero rows stay exactly zero."""
    from tests import trained_weights as TW
    lay = ML.layout(3)
    p = TW.weights(kind, 4)
    flat = dev(_flat(p, 3))
    n_rays, spr = 1024, 192
    P = n_rays * spr
    g = torch.Generator().manual_seed(18)
    pts = dev(torch.rand(P, 3, generator=g) * 2.4 - 1.2)
    vd = torch.randn(n_rays, 3, generator=g)
    vd = dev(vd / vd.norm(dim=-1, keepdim=True))
    d_raw = torch.randn(P, 4, generator=g) * 10.0 ** torch.randint(-30, 11, (P, 1), generator=g).float()
    d_raw[::97] = 0.0
    d_raw = dev(d_raw)
    rw = ops.pack_resident(flat)
    save = ops.save_workspace(P, "cuda")
    ops.mlp_fwd(pts, vd, spr, ops.pack_weights(flat, "fwd"), save, planes=rw)
    wb = ops.pack_weights(flat, "bwd")
    ga, pa, va = ops.mlp_bwd(d_raw, pts, vd, spr, wb, save)
    gb, pb, vb = ops.mlp_bwd(d_raw, pts, vd, spr, wb, save, planes=rw)
    Pp = ML.padded_samples(P)
    goff, _ = ML.section_offsets(ML.GRAD_SECTIONS, P)

    def rows(gr, name, width):
        return gr[goff[name]: goff[name] + width * Pp].view(Pp // 32, width // 32, 4, 2, 32, 4).permute(0, 4, 1, 2, 3, 5).reshape(Pp, width)[:P]
    zero = (d_raw.abs().sum(1) == 0)
    if kind == "adversarial":
        # The hand-made layers make the CHAIN ill-conditioned (pts_linears.3's +c / -c row meets pts_linears.2's identical row
        # pairs: terms 2^10 larger than their sum cancel in W_2^T dZ_2), so two correct fp32 evaluations differ from each other
        # by more than 2e-5 of a row.  The yardstick is fp64 instead: the same chain, on the same saved gates, in double
        # precision (torch on the device) -- the resident chain may be no further from it than the fused fp32 chain is.
        off, _ = ML.section_offsets(lay.save_sections, P)

        def srows(name, width=256):
            return save[off[name]: off[name] + width * Pp].view(Pp // 32, width // 32, 4, 2, 32, 4).permute(0, 4, 1, 2, 3, 5).reshape(Pp, width)[:P]
        W = lambda name: dev(p[name]).double()
        dr = d_raw.double()
        ref = {}
        ref["dzv"] = (dr[:, :3] @ W("rgb_linear.weight")) * (srows("hv", 128) > 0)
        ref["dfeat"] = (ref["dzv"] @ W("views_linears.0.weight"))[:, :256]
        d = (ref["dfeat"] @ W("feature_linear.weight") + dr[:, 3:4] * W("alpha_linear.weight")) * (srows("act7") > 0)
        ref["dz7"] = d
        for l in range(7, 0, -1):
            w_l = W("pts_linears.%d.weight" % l)
            if l == 5:
                w_l = w_l[:, lay.in_pts:]                       # (the skip layer's activation columns)
            d = (d @ w_l) * (srows("act%d" % (l - 1)) > 0)
            ref["dz%d" % (l - 1)] = d
        rep = {}
        for name, width in ML.GRAD_SECTIONS:
            a, b = rows(ga, name, width).double(), rows(gb, name, width).double()
            assert bool(torch.isfinite(b).all()), name
            size = ref[name].abs().max(1)[0].clamp_min(1e-300)
            ea, eb = ((a - ref[name]).abs().max(1)[0] / size)[~zero], ((b - ref[name]).abs().max(1)[0] / size)[~zero]
            rep[name] = {"fp32_max": float(ea.max()), "resident_max": float(eb.max()),
                         "fp32_q999": float(torch.quantile(ea[:1 << 20], 0.999)), "resident_q999": float(torch.quantile(eb[:1 << 20], 0.999))}
            assert rep[name]["resident_max"] <= 3.0 * rep[name]["fp32_max"] + 1e-7, (name, rep[name])
            assert rep[name]["resident_q999"] <= 2.0 * rep[name]["fp32_q999"] + 1e-7, (name, rep[name])
            assert bool((b[zero] == 0).all()), name
        from tests import parity_attribution as PA
        PA.REPORT["resident_data_gradient_chain_adversarial_weights_row_error_vs_fp64"] = rep
        return
    for name, width in ML.GRAD_SECTIONS:
        a, b = rows(ga, name, width).double(), rows(gb, name, width).double()
        assert bool(torch.isfinite(b).all()), name
        size = a.abs().max(1)[0]
        err = (a - b).abs().max(1)[0]
        assert bool((err <= 2e-5 * size + 1e-45).all()), (name, float((err / size.clamp_min(1e-300)).max()))
        assert bool((b[zero] == 0).all()), name
    for a, b, what in ((pa, pb, "d pts"), (va, vb, "d viewdirs")):
        size = a.double().abs().max(1)[0]
        err = (a.double() - b.double()).abs().max(1)[0]
        assert bool((err <= 1e-4 * size + 1e-45).all()), (what, float((err / size.clamp_min(1e-300)).max()))


@pytest.mark.parametrize("n,sf,train", [(4096, 128, True), (1025, 128, False), (777, 64, True), (300, 192, True)])
def test_fused_fine_stage_equals_the_three_launches(ops, n, sf, train):
    """scnerf_fine_stage_fwd_h3 -- the fine stage of render_rays (NeRF/render.py:269-285) as ONE launch: inverse-cdf sampler
    and merge in front of the resident network, compositing behind, whole rays through passes of 128 samples -- against
    fine_sample -> mlp_fwd (resident) -> composite_fwd: the same device code on the same numbers, so every output is
    bit-identical: depths, search indices, cdf, points, raw, maps, weights, the activation workspace and the chunk maxima."""
    from tests.emu_mlp_util import network_params
    sc, tot = 64, 64 + sf
    flat = dev(_flat(network_params(4, 3), 3))
    wf, rw = ops.pack_weights(flat, "fwd"), ops.pack_resident(flat)
    rays = synth.ray_batch(n, seed=4).cuda()
    g = torch.Generator().manual_seed(9)
    z_c = torch.sort(torch.rand(n, sc, generator=g), -1)[0].cuda()
    w_c = (torch.rand(n, sc, generator=g) ** 4)
    w_c[::7, 10:50] = 0.0                                          # empty bins: the `denom < 1e-5` branch
    w_c = w_c.cuda()
    u = torch.rand(n, sf, generator=g).cuda()
    noise = (torch.randn(n, tot, generator=g) * 0.5).cuda()
    P = n * tot
    save_a = ops.save_workspace(P, "cuda").fill_(float("nan")) if train else None
    save_b = ops.save_workspace(P, "cuda").fill_(float("nan")) if train else None
    mx_a, mx_b = (ops.ChunkMaxima(P, "cuda"), ops.ChunkMaxima(P, "cuda")) if train else (None, None)
    z_f, pts_f, z_s, z_std, inds, cdf = ops.fine_sample(rays, z_c, w_c, u, True, True)
    raw = ops.mlp_fwd(pts_f, rays[:, 8:11], tot, wf, save_a, planes=rw, maxima=mx_a).view(n, tot, 4)
    rgb, disp, acc, w, depth = ops.composite_fwd(raw, z_f, rays, noise, True)
    got = ops.fine_stage_fwd(rays, z_c, w_c, u, wf, save_b, noise, True, rw, maxima=mx_b, want_inds=True, want_cdf=True,
                             want_weights=True)
    want = (z_f, pts_f, z_s, z_std, inds, cdf, raw, rgb, disp, acc, depth, w)
    names = ("z_f", "pts_f", "z_samples", "z_std", "inds", "cdf", "raw", "rgb", "disp", "acc", "depth", "weights")
    for name, a, b in zip(names, want, got):
        assert torch.equal(a, b), (name, n, sf)
    if train:
        wrote = ~torch.isnan(save_a)
        assert torch.equal(wrote, ~torch.isnan(save_b))
        assert torch.equal(save_a[wrote].view(torch.int32), save_b[wrote].view(torch.int32))
        assert torch.equal(mx_a.x, mx_b.x)
