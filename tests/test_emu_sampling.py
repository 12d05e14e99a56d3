"""Logic tests of the sampling kernels: the HIP sources run under the CPU SIMT
interpreter (tests/emu) and are compared with the golden vectors of the reference and
with the oracle.  Indices and the cdf are compared bit-for-bit."""
import numpy as np
import pytest
import torch

from oracle import scnerf_oracle as O
from scnerf_amd import synthetic as synth
from tests.emu import harness as H
from conftest import t

pytestmark = pytest.mark.emu


def run_sample_pdf(bins, w, u):
    n, nb = bins.shape
    ns = u.shape[-1]
    stride = ns if u.ndim == 2 else 0
    samples = np.zeros((n, ns), np.float32)
    inds = np.zeros((n, ns), np.int64)
    cdf = np.zeros((n, nb), np.float32)
    H.call("scnerf_sample_pdf", H.f32(bins), H.f32(w), H.f32(u), stride, samples, inds, cdf, n, nb, ns, None)
    return samples, inds, cdf


@pytest.mark.parametrize("tag", ["rand", "knot", "det"])
def test_sample_pdf_golden_bit_exact(golden, tag):
    g = golden("sample_pdf")
    u = g[tag + "/u"]
    if tag == "det":
        u = np.ascontiguousarray(u[0])          # one shared linspace row (u_row_stride = 0)
    samples, inds, cdf = run_sample_pdf(g["bins"], g["weights"], u)
    np.testing.assert_array_equal(cdf, g[tag + "/cdf"])
    np.testing.assert_array_equal(inds, g[tag + "/inds"])
    np.testing.assert_array_equal(samples, g[tag + "/samples"])


@pytest.mark.parametrize("nb,ns", [(3, 1), (9, 5), (31, 64), (63, 128), (64, 200), (127, 96), (200, 33),
                                   (601, 40), (1100, 16)])      # >= 512 weights: the row sum's cascade levels engage
def test_sample_pdf_vs_oracle_shapes(nb, ns):
    g = torch.Generator().manual_seed(nb * 1000 + ns)
    n = 7          # ragged against the 4-rays-per-block tiling
    bins = torch.sort(torch.rand(n, nb, generator=g), -1)[0]
    w = torch.rand(n, nb - 1, generator=g) ** 3
    u = torch.rand(n, ns, generator=g)
    s, inds, cdf = run_sample_pdf(bins.numpy(), w.numpy(), u.numpy())
    if 8 <= nb - 1 < 512:
        so, io, co = O.sample_pdf(bins, w, u, rowsum="aten")
    else:
        so, io, co = O.sample_pdf(bins, w, u)
    np.testing.assert_array_equal(cdf, co.numpy())
    np.testing.assert_array_equal(inds, io.numpy())
    np.testing.assert_array_equal(s, so.numpy())


@pytest.mark.parametrize("side", ["left", "right"])
@pytest.mark.parametrize("Ba,Bv,A,V", [(1, 1, 1, 1), (1, 5, 50, 12), (5, 1, 50, 12), (6, 6, 1, 12),
                                       (6, 6, 63, 128), (3, 3, 500, 120), (9, 9, 64, 1)])
def test_searchsorted_matches_numpy(side, Ba, Bv, A, V):
    """Re-points the reference's only KAT-style test (NeRF/torchsearchsorted/test/
    test_searchsorted.py:9-44: row-wise np.searchsorted oracle, broadcast rows, both sides)
    at the HIP search."""
    rng = np.random.default_rng(Ba * 7 + Bv * 3 + A + V)
    a = np.sort(rng.random((Ba, A), dtype=np.float32), -1)
    v = rng.random((Bv, V), dtype=np.float32)
    v[:, ::3] = a[:1, rng.integers(0, A, size=v[:, ::3].shape[1])] if Ba >= 1 else v[:, ::3]  # exact ties
    nrow = max(Ba, Bv)
    out = np.zeros((nrow, V), np.int64)
    H.call("scnerf_searchsorted", a, v, out, nrow, Ba, Bv, A, V, int(side == "left"), None)
    for r in range(nrow):
        exp = np.searchsorted(a[0 if Ba == 1 else r], v[0 if Bv == 1 else r], side=side)
        np.testing.assert_array_equal(out[r], exp)
    assert out.dtype == np.int64


def test_searchsorted_survey_kat():
    cdf = np.array([[0.0, 0.2, 0.2, 0.7, 1.0]], np.float32)
    u = np.array([[0.0, 0.2, 0.5, 1.0]], np.float32)
    out = np.zeros((1, 4), np.int64)
    H.call("scnerf_searchsorted", cdf, u, out, 1, 1, 1, 5, 4, 0, None)
    assert out.tolist() == [[1, 3, 3, 5]]


@pytest.mark.parametrize("lindisp", [0, 1])
@pytest.mark.parametrize("perturb", [0, 1])
def test_coarse_sample_bit_exact(lindisp, perturb):
    n, s = 37, 64
    rays = synth.ray_batch(n, seed=5, lindisp=bool(lindisp))
    t_rand = synth.render_randoms(n, s, 0, seed=6)["t_rand"]
    t_vals = torch.linspace(0.0, 1.0, steps=s)
    z = np.zeros((n, s), np.float32)
    pts = np.zeros((n, s, 3), np.float32)
    H.call("scnerf_coarse_sample", rays.numpy(), 11, t_vals.numpy(), t_rand.numpy() if perturb else None,
           z, pts, n, s, lindisp, None)
    zo = O.stratified_z(rays[:, 6:7], rays[:, 7:8], s, bool(lindisp), t_rand if perturb else None)
    po = rays[:, None, 0:3] + rays[:, None, 3:6] * zo[:, :, None]
    np.testing.assert_array_equal(z, zo.numpy())
    np.testing.assert_array_equal(pts, po.numpy())


@pytest.mark.parametrize("sc,sf,det", [(64, 128, False), (64, 64, False), (64, 128, True), (16, 24, False), (64, 200, False), (64, 192, False)])
def test_fine_sample_bit_exact(sc, sf, det):
    n = 10
    g = torch.Generator().manual_seed(sc + sf)
    rays = synth.ray_batch(n, seed=7)
    z_c = O.stratified_z(rays[:, 6:7], rays[:, 7:8], sc, False, torch.rand(n, sc, generator=g))
    w_c = torch.rand(n, sc, generator=g) ** 5
    w_c[1] = 0.0                     # empty ray: flat 1e-5 pdf
    w_c[2, : sc // 2] = 0.0
    u = O.deterministic_u(n, sf) if det else torch.rand(n, sf, generator=g)
    tot = sc + sf
    z_f = np.zeros((n, tot), np.float32)
    pts_f = np.zeros((n, tot, 3), np.float32)
    z_s = np.zeros((n, sf), np.float32)
    z_std = np.zeros((n,), np.float32)
    inds = np.zeros((n, sf), np.int64)
    cdf = np.zeros((n, sc - 1), np.float32)
    uu = np.ascontiguousarray(u[0].numpy()) if det else u.contiguous().numpy()
    H.call("scnerf_fine_sample", rays.numpy(), 11, z_c.numpy(), w_c.numpy(), uu, 0 if det else sf,
           z_f, pts_f, z_s, z_std, inds, cdf, n, sc, sf, None)
    z_mid = 0.5 * (z_c[:, 1:] + z_c[:, :-1])
    so, io, co = O.sample_pdf(z_mid, w_c[:, 1:-1], u.contiguous(), rowsum="aten" if sc - 2 >= 8 else "torch")
    zf_o = torch.sort(torch.cat([z_c, so], -1), -1)[0]
    pts_o = rays[:, None, 0:3] + rays[:, None, 3:6] * zf_o[:, :, None]
    np.testing.assert_array_equal(cdf, co.numpy())
    np.testing.assert_array_equal(inds, io.numpy())
    np.testing.assert_array_equal(z_s, so.numpy())
    np.testing.assert_array_equal(z_f, zf_o.numpy())
    np.testing.assert_array_equal(pts_f, pts_o.numpy())
    np.testing.assert_allclose(z_std, torch.std(so, dim=-1, unbiased=False).numpy(), rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("sc,sf", [(64, 128), (40, 33), (64, 200)])
def test_fine_sample_merge_is_torch_sort_with_ties_and_nans(sc, sf):
    """The rank merge of the fine sampler (ballot counts over register-held values up to 256 depths, the pairwise walk beyond)
    against torch.sort of the concatenation [z_c | new samples] on rows with repeated depths, +-0, infinities and NaNs
    (NaN last, ties by position): the merged row bit for bit, the new samples taken from the kernel's own output."""
    n = 6
    g = torch.Generator().manual_seed(sc * 7 + sf)
    rays = synth.ray_batch(n, seed=3)
    z_c = torch.sort(torch.rand(n, sc, generator=g), -1)[0]
    z_c[0, 5:9] = z_c[0, 5]                         # repeated coarse depths
    z_c[1, 0] = 0.0
    z_c[1, 1] = -0.0                                # equal in value, different in bits
    z_c[2, -1] = float("inf")
    z_c[3, 10] = float("nan")                       # (its bins are NaN too: NaN samples, several of them)
    z_c[4] = 0.25                                   # every depth the same
    w_c = torch.rand(n, sc, generator=g) ** 3
    w_c[5, 2:-2] = 0.0                              # flat cdf: many samples on one spot
    u = torch.rand(n, sf, generator=g)
    u[5, : sf // 2] = 0.5
    tot = sc + sf
    z_f = np.zeros((n, tot), np.float32)
    pts_f = np.zeros((n, tot, 3), np.float32)
    z_s = np.zeros((n, sf), np.float32)
    z_std = np.zeros((n,), np.float32)
    H.call("scnerf_fine_sample", rays.numpy(), 11, z_c.numpy(), w_c.numpy(), u.numpy(), sf, z_f, pts_f, z_s, z_std, None, None, n, sc, sf, None)
    want = torch.sort(torch.cat([z_c, torch.from_numpy(z_s)], -1), dim=-1, stable=True)[0].numpy()
    np.testing.assert_array_equal(z_f.view(np.int32), want.view(np.int32))
