"""The resident arithmetic (csrc/mlp_h3.h: three fp16 products, register-resident activations, one launch) on the CPU
SIMT interpreter against the oracle network and against the fused fp32 kernels' workspaces."""
import numpy as np
import pytest
import torch

from oracle import scnerf_oracle as O
from scnerf_amd import mlp_layout as ML
from tests.emu import harness as H
from tests.emu_mlp_util import network_params, pack_forward, pack_h3, save_views, oracle_activations, flat_params

pytestmark = pytest.mark.emu


@pytest.mark.parametrize("pd", [3, 4])
def test_stream_tables_reference_every_weight_twice(pd):
    lay = ML.layout(pd)
    for kind in ("fwd", "bwd"):
        idx, meta, n, parts = ML.h3_plan(pd, kind)
        assert idx.shape[0] == meta.shape[0] * 512 and meta.shape[0] % ML.H3_CHUNK_FRAGS == 0
        assert sum(u for _, u in parts) * 4 == n
        cnt = np.bincount(idx[idx >= 0], minlength=lay.n_params)
        po = lay.param_offsets
        for name, shape in lay.param_shapes:
            sl = slice(po[name], po[name] + int(np.prod(shape)))
            want = 2 if (name.endswith(".weight") and not name.startswith("alpha")) else 0
            assert cnt[sl].min() == want and cnt[sl].max() == want, (kind, name)


def test_packed_planes_reassemble_the_weights():
    p = network_params(0)
    fwd, bwd, sc = pack_h3(p)
    src = flat_params(p)
    sc = sc[:96].reshape(12, 8)
    jobs = ML.h3_scale_jobs(3)
    for l in range(12):
        w = src[jobs[l, 0]: jobs[l, 0] + jobs[l, 1] * jobs[l, 2]].reshape(jobs[l, 1], jobs[l, 2])
        b = src[jobs[l, 3]: jobs[l, 3] + jobs[l, 1]]
        assert sc[l, 0] * sc[l, 1] == 1.0 and np.log2(sc[l, 0]) == np.round(np.log2(sc[l, 0]))
        assert 2 ** 12 <= np.abs(w).max() * sc[l, 0] < 2 ** 13
        assert np.abs(w).sum(1).max() <= sc[l, 2] <= np.abs(w).sum(1).max() * 1.001
        assert sc[l, 3] == np.abs(b).max()
        assert np.abs(w).sum(0).max() <= sc[l, 4] <= np.abs(w).sum(0).max() * 1.001
    for kind, stream in (("fwd", fwd), ("bwd", bwd)):
        idx, meta, n, _ = ML.h3_plan(3, kind)
        vals = stream.view(np.float16).astype(np.float64).reshape(-1, 512)
        idx = idx.reshape(-1, 512)
        acc = np.zeros(src.shape[0])
        for f in range(n):
            ok = idx[f] >= 0
            np.add.at(acc, idx[f][ok], vals[f][ok] / sc[meta[f] >> 1, 0])
            assert np.all(vals[f][~ok] == 0)
        used = np.zeros(src.shape[0], bool)
        used[idx[idx >= 0]] = True
        rel = np.abs(acc[used] - src[used]) / np.maximum(np.abs(src[used]), 1e-30)
        big = np.abs(src[used]) * 1.0 > 0           # every weight: h + l within 2^-21 of the layer's largest
        assert big.all()
        lim = np.zeros(src.shape[0])
        for l in range(12):
            lim[jobs[l, 0]: jobs[l, 0] + jobs[l, 1] * jobs[l, 2]] = 2.0 ** -21 * sc[l, 5]
        assert np.all(np.abs(acc[used] - src[used]) <= lim[used])


@pytest.mark.parametrize("pd", [3, 4])
@pytest.mark.parametrize("n_rays,spr,save", [(5, 32, True), (3, 64, False)])
def test_resident_forward_matches_oracle(n_rays, spr, save, pd):
    lay = ML.layout(pd)
    IN = lay.in_pts
    p = network_params(0 if pd == 3 else 777, pd)
    wpk = pack_forward(p, pd)
    fwd, _, sc = pack_h3(p, pd, directions=("fwd",))
    P = n_rays * spr
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand(P, pd, generator=g) * 3 - 1.5)
    vd = torch.randn(n_rays, 3, generator=g)
    vd = vd / vd.norm(dim=-1, keepdim=True)
    raw = np.full((P, 4), np.nan, np.float32)
    sv = np.full(lay.save_floats(P), np.nan, np.float32) if save else None
    H.call("scnerf_mlp_fwd_h3", pd, pts.numpy(), vd.numpy(), 3, spr, wpk, fwd, sc, raw, sv, P, None, 0, 0, None)
    ref = O.query_network(p, pts.reshape(n_rays, spr, pd), vd).reshape(P, 4)
    np.testing.assert_allclose(raw, ref.numpy(), rtol=2e-5, atol=2e-5)
    if save:
        vps = vd[:, None, :].expand(n_rays, spr, 3).reshape(P, 3)
        oa = oracle_activations(p, pts, vps)
        s = save_views(sv, P, pd)
        np.testing.assert_allclose(s["epts"][:, :IN], oa["e"].numpy(), rtol=0, atol=2e-6)
        assert np.all(s["epts"][:, IN:] == 0)
        np.testing.assert_allclose(s["eviews"][:, :27], oa["ev"].numpy(), rtol=0, atol=2e-6)
        for l in range(8):
            np.testing.assert_allclose(s["act%d" % l], oa["acts"][l].numpy(), rtol=2e-5, atol=2e-5, err_msg="act%d" % l)
        np.testing.assert_allclose(s["feat"], oa["feat"].numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(s["hv"], oa["hv"].numpy(), rtol=2e-5, atol=2e-5)
        for sec, got, ntile in [(l, s["act%d" % l], 8) for l in range(8)] + [(8, s["hv"], 4)]:
            for p_ in (0, 31, 77, P - 1):
                wt, m = divmod(p_, 32)
                for hh in (0, 1):
                    words = s["mask"][sec, wt, m + 32 * hh]
                    for t_ in range(ntile):
                        for r in range(16):
                            i = 16 * t_ + r
                            bit = (int(words[i >> 5]) >> (31 - (i & 31))) & 1
                            assert bit == int(got[p_, ML.feat_of(t_, r, hh)] > 0)


@pytest.mark.parametrize("pd", [3, 4])
@pytest.mark.parametrize("n_rays,spr", [(5, 32), (1, 70)])
def test_resident_dgrad_matches_autograd(n_rays, spr, pd):
    from tests.test_emu_mlp_bwd import oracle_backward
    from tests.emu_mlp_util import pack_backward, grad_views
    lay = ML.layout(pd)
    p = network_params(2 if pd == 3 else 778, pd)
    wpk, wbk = pack_forward(p, pd), pack_backward(p, pd)
    fwd, bwd, sc = pack_h3(p, pd)
    P = n_rays * spr
    g = torch.Generator().manual_seed(9)
    pts = torch.rand(P, pd, generator=g) * 2.4 - 1.2
    vd = torch.randn(n_rays, 3, generator=g)
    vd = vd / vd.norm(dim=-1, keepdim=True)
    d_raw = torch.randn(P, 4, generator=g)
    # per-sample magnitudes over many orders: the scales are per sample
    d_raw = d_raw * (10.0 ** torch.randint(-12, 4, (P, 1), generator=g).float())
    d_raw[3] = 0.0
    raw = np.zeros((P, 4), np.float32)
    save = np.full(lay.save_floats(P), np.nan, np.float32)
    H.call("scnerf_mlp_fwd_h3", pd, pts.numpy(), vd.numpy(), 3, spr, wpk, fwd, sc, raw, save, P, None, 0, 0, None)
    grads = np.full(ML.grad_floats(P), np.nan, np.float32)
    d_pts = np.full((P, pd), np.nan, np.float32)
    d_views = np.full((P, 3), np.nan, np.float32)
    H.call("scnerf_mlp_bwd_h3", pd, d_raw.numpy(), pts.numpy(), vd.numpy(), 3, spr, wbk, bwd, sc, save, grads, d_pts, d_views, P, None, 0, 0, None)
    ref = oracle_backward(p, pts, vd, spr, d_raw)
    gv = grad_views(grads, P)

    def close(a, b, what):
        # row by row: every sample against its own size
        scale = np.abs(b).reshape(P, -1).max(1, keepdims=True) + 1e-30
        err = np.abs(a - b).reshape(P, -1)
        bad = err > 3e-5 * scale + 1e-37
        # (a ReLU gate the two sides decide differently changes whole rows: such rows are rare and checked loosely)
        rows = bad.any(1)
        assert rows.mean() <= 0.02, "%s: %d of %d rows off (worst %g of its row)" % (what, rows.sum(), P, float((err / scale).max()))

    close(gv["dzv"], ref["dzv"].numpy(), "dzv")
    close(gv["dfeat"], ref["dfeat"].numpy(), "dfeat")
    for l in range(7, -1, -1):
        close(gv["dz%d" % l], ref["dz"][l].numpy(), "dz%d" % l)
    close(d_pts, ref["d_pts"].numpy(), "d_pts")
    assert np.all(gv["dz0"][3] == 0) and np.all(d_pts[3] == 0)
    got_vd = d_views.reshape(n_rays, spr, 3).sum(1)
    sv = float(np.abs(ref["d_vd"].numpy()).max())
    assert float(np.abs(got_vd - ref["d_vd"].numpy()).max()) <= 3e-5 * sv


def _tiled(m):
    """[P, 256] -> tile-native section (P a multiple of 32)"""
    P = m.shape[0]
    return np.ascontiguousarray(m.reshape(P // 32, 32, 8, 4, 2, 4).transpose(0, 2, 3, 4, 1, 5)).reshape(-1)


def test_half_weight_gradient_gemm_is_fp32_grade_on_the_interpreter():
    """The 256 x 256 weight-gradient GEMM on three fp16 products with one scale per operand and workgroup chunk
    (csrc/wgrad256_half.h) against fp64: per-sample gradient magnitudes spread over 2^40, half of X zero (post-ReLU), an
    all-zero chunk.  With magnitudes this far apart a handful of samples carries each sum, so the bound is that of a
    SINGLE product of two cut operands -- 3 x 2^-22 of |dz x| (two cuts and the dropped low x low term) -- not the
    averaged-out error of a long sum, which the GPU test measures against the fp32 kernel on 65 536 samples.  Bias sums
    exact to fp32 summation."""
    rng = np.random.default_rng(5)
    P, chunks = 512, 4                       # 128 samples per chunk = 8 slabs
    dz = rng.standard_normal((P, 256)).astype(np.float32) * (2.0 ** rng.integers(-30, 10, (P, 1))).astype(np.float32)
    x = rng.standard_normal((P, 256)).astype(np.float32) * (2.0 ** rng.integers(-3, 3, (P, 256))).astype(np.float32)
    x *= rng.random((P, 256)) < 0.5
    dz[128:256] = 0.0                        # a chunk whose gradient vanishes
    chunk = H.lib().scnerf_wgrad_chunk_samples(P, chunks)
    assert chunk == 128
    amax_dz = np.abs(dz).reshape(chunks, -1).max(1).astype(np.float32)
    amax_x = np.abs(x).reshape(chunks, -1).max(1).astype(np.float32)
    ws = np.full(chunks * (65536 + 256), np.nan, np.float32)
    dW = np.full((256, 256), np.nan, np.float32)
    db = np.full(256, np.nan, np.float32)
    H.call("scnerf_wgrad256_half", _tiled(dz), _tiled(x), P, chunks, ws, dW, db, amax_dz, amax_x, None)
    ref = dz.astype(np.float64).T @ x.astype(np.float64)
    scale = np.abs(dz).astype(np.float64).T @ np.abs(x).astype(np.float64)
    err = np.abs(dW - ref) / scale
    assert err.max() < 3 * 2.0 ** -22 and np.sqrt((err * err).mean()) < 1e-7, (err.max(), np.sqrt((err * err).mean()))
    assert (np.abs(db - dz.astype(np.float64).sum(0)) / np.abs(dz).astype(np.float64).sum(0)).max() < 1e-6     # fp32 sums
    # maxima given too LARGE (a chunk mate outside this GEMM's view) only cost low-order bits of the small values
    dW2 = np.full((256, 256), np.nan, np.float32)
    H.call("scnerf_wgrad256_half", _tiled(dz), _tiled(x), P, chunks, ws, dW2, db, amax_dz * 64, amax_x * 64, None)
    assert (np.abs(dW2 - ref) / scale).max() < 3 * 2.0 ** -22


def _tiled_w(m):
    """[P, W] -> tile-native section of width W (P a multiple of 32, W of 32)"""
    P, W = m.shape
    return np.ascontiguousarray(m.reshape(P // 32, 32, W // 32, 4, 2, 4).transpose(0, 2, 3, 4, 1, 5)).reshape(-1)


@pytest.mark.parametrize("shape", ["256x64", "256x128", "128x256"])
def test_narrow_half_weight_gradient_gemms_on_the_interpreter(shape):
    """The narrow weight-gradient GEMMs on three fp16 products (csrc/wgrad_half_narrow.h) against fp64: a row-major X
    with garbage in the rows beyond the valid samples (staged as zeros), valid columns fewer than loaded ones, more
    workgroups than coarse chunks of the maxima (a workgroup takes the largest maximum its range touches), a workgroup
    range straddling two coarse chunks, an all-zero stretch; bias sums exact to fp32 summation."""
    rng = np.random.default_rng(11)
    wa, wb = (int(v) for v in shape.split("x"))
    tiled_x = wb == 256
    k_out = wb if tiled_x else wb - 1
    P, Ppad, chunks = 500, 512, 8                      # 64 samples per workgroup = 4 slabs
    n_coarse, coarse = 3, 192                          # 192, 192, 128 samples: workgroup 2 and 5 straddle
    dz = np.zeros((Ppad, wa), np.float32)
    dz[:P] = rng.standard_normal((P, wa)).astype(np.float32) * (2.0 ** rng.integers(-20, 6, (P, 1))).astype(np.float32)
    dz[256:320] = 0.0
    x = rng.standard_normal((Ppad, wb)).astype(np.float32) * (2.0 ** rng.integers(-3, 3, (Ppad, wb))).astype(np.float32)
    if tiled_x:
        x[P:] = 0.0                                    # a tile-native X is zero beyond the valid samples (the forward writes it)
        xin = _tiled_w(x)
    else:
        x[P:] = np.nan                                 # a row-major X may hold anything there
        xin = x.reshape(-1).copy()
    xv = np.where(np.arange(Ppad)[:, None] < P, x, 0.0)
    amax_dz = np.array([np.abs(dz[c * coarse:(c + 1) * coarse]).max() for c in range(n_coarse)], np.float32)
    amax_x = np.array([np.abs(xv[c * coarse:(c + 1) * coarse]).max() for c in range(n_coarse)], np.float32)
    ws = np.full(H.lib().scnerf_wgrad_workspace_floats(wa, wb, chunks), np.nan, np.float32)
    dW = np.full((wa, k_out), np.nan, np.float32)
    db = np.full(wa, np.nan, np.float32)
    H.call("scnerf_wgrad_half_narrow", _tiled_w(dz), wa, xin, wb, k_out, int(tiled_x), P, chunks, ws, dW, db, amax_dz, amax_x,
           n_coarse, coarse, None)
    ref = dz.astype(np.float64).T @ xv.astype(np.float64)[:, :k_out]
    scale = np.abs(dz).astype(np.float64).T @ np.abs(xv).astype(np.float64)[:, :k_out]
    err = np.abs(dW - ref) / scale
    assert err.max() < 3 * 2.0 ** -22 and np.sqrt((err * err).mean()) < 1e-7, (err.max(), np.sqrt((err * err).mean()))
    assert (np.abs(db - dz.astype(np.float64).sum(0)) / np.abs(dz).astype(np.float64).sum(0)).max() < 1e-6


@pytest.mark.parametrize("where,value", [("pts_linears.3.weight", float("nan")), ("views_linears.0.bias", float("inf")),
                                         ("alpha_linear.weight", float("nan"))])
def test_a_parameter_that_is_not_finite_shows_in_the_colours(where, value):
    """The reference carries a NaN parameter through torch.relu and every matrix product to the loss
    (NeRF/run_nerf_helpers.py:105-128).  The resident kernels' ReLU and maxima are v_max-based and drop NaNs, so the scale
    pass flags a network with a NaN / infinite weight or bias and the forward makes every colour of the launch NaN: a
    diverged run cannot produce finite-looking renders.  A finite network leaves the flag at exactly 0."""
    p = network_params(0)
    wpk = pack_forward(p, 3)
    fwd, _, sc = pack_h3(p, 3, directions=("fwd",))
    assert sc[:96].reshape(12, 8)[10, 7] == 0.0 and np.isfinite(sc[:96]).all()
    bad = {k: v.clone() for k, v in p.items()}
    bad[where].reshape(-1)[5] = value
    fwd_b, _, sc_b = pack_h3(bad, 3, directions=("fwd",))
    assert np.isnan(sc_b[:96].reshape(12, 8)[10, 7])
    P = 64
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand(P, 3, generator=g) * 3 - 1.5)
    vd = torch.nn.functional.normalize(torch.randn(2, 3, generator=g), dim=-1)
    raw = np.zeros((P, 4), np.float32)
    H.call("scnerf_mlp_fwd_h3", 3, pts.numpy(), vd.numpy(), 3, 32, pack_forward(bad, 3), fwd_b, sc_b, raw, None, P, None, 0, 0, None)
    assert np.isnan(raw[:, :3]).all()
    raw_ok = np.zeros((P, 4), np.float32)
    H.call("scnerf_mlp_fwd_h3", 3, pts.numpy(), vd.numpy(), 3, 32, wpk, fwd, sc, raw_ok, None, P, None, 0, 0, None)
    assert np.isfinite(raw_ok).all()


@pytest.mark.parametrize("n,sf,train,det", [(3, 128, True, False), (2, 64, False, False), (1, 192, True, True)])
def test_fused_fine_stage_equals_the_three_launches(n, sf, train, det):
    """scnerf_fine_stage_fwd_h3 (sampler + merge in front of the resident network, compositing behind, whole rays through
    passes of 128 samples; NeRF/render.py:269-285) against scnerf_fine_sample -> scnerf_mlp_fwd_h3 -> scnerf_composite_fwd on
    the same inputs: the same device code on the same numbers -- every output BIT-identical, including the search indices,
    the activation workspace and (odd ray counts) the last workgroup's half-empty pass."""
    from scnerf_amd import synthetic as synth
    sc, tot = 64, 64 + sf
    p = network_params(1)
    wpk = pack_forward(p, 3)
    fwd, _, scales = pack_h3(p, 3, directions=("fwd",))
    rays = synth.ray_batch(n, seed=4).numpy()
    g = torch.Generator().manual_seed(9)
    z_c = torch.sort(torch.rand(n, sc, generator=g), -1)[0].numpy()
    w_c = (torch.rand(n, sc, generator=g) ** 4).numpy()
    w_c[0, 10:50] = 0.0                                            # empty bins: the `denom < 1e-5` branch
    u = torch.linspace(0, 1, sf).numpy() if det else torch.rand(n, sf, generator=g).numpy()
    noise = (torch.randn(n, tot, generator=g) * 0.5).numpy()
    lay = ML.layout(3)
    P = n * tot

    def outputs():
        return dict(z_f=np.full((n, tot), np.nan, np.float32), pts=np.full((n, tot, 3), np.nan, np.float32),
                    z_s=np.full((n, sf), np.nan, np.float32), z_std=np.full(n, np.nan, np.float32),
                    inds=np.full((n, sf), -1, np.int64), cdf=np.full((n, sc - 1), np.nan, np.float32),
                    raw=np.full((n, tot, 4), np.nan, np.float32), rgb=np.full((n, 3), np.nan, np.float32),
                    disp=np.full(n, np.nan, np.float32), acc=np.full(n, np.nan, np.float32), depth=np.full(n, np.nan, np.float32),
                    w=np.full((n, tot), np.nan, np.float32), save=np.full(lay.save_floats(P), np.nan, np.float32) if train else None)
    a, b = outputs(), outputs()
    stride = 0 if det else sf
    H.call("scnerf_fine_sample", rays, 11, z_c, w_c, u, stride, a["z_f"], a["pts"], a["z_s"], a["z_std"], a["inds"], a["cdf"], n, sc, sf, None)
    vd = np.ascontiguousarray(rays[:, 8:11])
    H.call("scnerf_mlp_fwd_h3", 3, a["pts"], vd, 3, tot, wpk, fwd, scales, a["raw"], a["save"], P, None, 0, 0, None)
    H.call("scnerf_composite_fwd", a["raw"], a["z_f"], rays, 11, noise, 1, a["rgb"], a["disp"], a["acc"], a["depth"], a["w"], n, tot, None)
    H.call("scnerf_fine_stage_fwd_h3", rays, 11, z_c, w_c, u, stride, wpk, fwd, scales, b["save"], noise, 1, b["z_f"], b["pts"],
           b["z_s"], b["z_std"], b["inds"], b["cdf"], b["raw"], b["rgb"], b["disp"], b["acc"], b["depth"], b["w"], n, sc, sf,
           None, 0, 0, None)
    for k in a:
        if a[k] is None:
            continue
        x, y = a[k], b[k]
        if k == "save":                                             # (padding slots of the last wave tile are never written)
            ok = ~np.isnan(x)
            assert np.array_equal(np.isnan(x), np.isnan(y)), k
            x, y = x[ok], y[ok]
        assert not np.isnan(y).any(), k
        np.testing.assert_array_equal(x.view(np.int32) if x.dtype == np.float32 else x, y.view(np.int32) if y.dtype == np.float32 else y, err_msg=k)
    # shapes the fused stage does not cover are refused, not approximated
    assert H.lib_call_status("scnerf_fine_stage_fwd_h3", rays, 11, z_c, w_c, u, stride, wpk, fwd, scales, None, noise, 1, b["z_f"], b["pts"],
                             b["z_s"], b["z_std"], None, None, b["raw"], b["rgb"], b["disp"], b["acc"], None, None, n, sc, 100,
                             None, 0, 0, None) != 0
