"""Weight-gradient GEMM (HIP source under the CPU SIMT interpreter) vs numpy."""
import numpy as np
import pytest

from tests.emu import harness as H

pytestmark = pytest.mark.emu


def tile(mat, width):
    """row-major [P, width] -> tile-native flat section (padded to 128 samples, dead slots = 0)."""
    P = mat.shape[0]
    Pp = (P + 127) // 128 * 128
    full = np.zeros((Pp, width), np.float32)
    full[:P] = mat
    a = full.reshape(Pp // 32, 32, width // 32, 4, 2, 4)         # tile, m, t, q, h, j
    return np.ascontiguousarray(a.transpose(0, 2, 3, 4, 1, 5)).reshape(-1)   # tile, t, q, h, m, j


def test_tile_helper_is_the_inverse_of_untile():
    from scnerf_amd import mlp_layout as ML
    m = np.random.default_rng(0).standard_normal((77, 128)).astype(np.float32)
    np.testing.assert_array_equal(ML.untile(tile(m, 128), 128, 77), m)


@pytest.mark.parametrize("P,lda,n_load,n_out,a_tiled,ldb,k_load,k_out,b_tiled,chunks,ldo,col0", [
    (100, 256, 256, 256, 1, 256, 256, 256, 1, 3, 256, 0),     # trunk layer / feature
    (77, 256, 256, 256, 1, 64, 64, 63, 0, 2, 319, 0),        # layer 0 / skip part: encoded points (row-major)
    (50, 256, 256, 256, 1, 256, 256, 256, 1, 1, 319, 63),    # skip layer, h part at column offset 63
    (64, 128, 128, 128, 1, 256, 256, 256, 1, 2, 283, 0),     # views layer, feature part
    (33, 128, 128, 128, 1, 32, 32, 27, 0, 4, 283, 256),      # views layer, encoded direction part
    (90, 4, 4, 3, 0, 128, 128, 128, 1, 2, 128, 0),           # rgb layer from d_raw (row-major) x hv (tiled)
    (5, 256, 256, 256, 1, 256, 256, 256, 1, 7, 256, 0),       # more chunks than stages
    (300, 256, 256, 256, 0, 256, 256, 256, 0, 3, 256, 0),    # both operands row-major
])
@pytest.mark.parametrize("arith", [1, 0], ids=["split", "fp32"])
def test_wgrad(arith, P, lda, n_load, n_out, a_tiled, ldb, k_load, k_out, b_tiled, chunks, ldo, col0):
    if arith == 0 and not (a_tiled and b_tiled and n_load == 256 and k_load == 256):
        pytest.skip("the arithmetic switch only concerns the tile-native 256 x 256 GEMMs")
    assert H.lib().scnerf_wgrad_arithmetic(arith) == arith
    try:
        _wgrad_case(P, lda, n_load, n_out, a_tiled, ldb, k_load, k_out, b_tiled, chunks, ldo, col0)
    finally:
        assert H.lib().scnerf_wgrad_arithmetic(2) == 2


def _wgrad_case(P, lda, n_load, n_out, a_tiled, ldb, k_load, k_out, b_tiled, chunks, ldo, col0):
    rng = np.random.default_rng(P + n_load + k_load)
    A = rng.standard_normal((P, lda), dtype=np.float32)
    B = rng.standard_normal((P, ldb), dtype=np.float32)
    ws_n = H.lib().scnerf_wgrad_workspace_floats(n_load, k_load, chunks)
    assert ws_n > 0
    ws = np.full(ws_n, np.nan, np.float32)
    dW = np.full((n_out, ldo), np.nan, np.float32)
    db = np.full(n_out, np.nan, np.float32)
    Ad = tile(A, lda) if a_tiled else A
    Bd = tile(B, ldb) if b_tiled else B
    H.call("scnerf_wgrad", Ad, lda, n_load, n_out, a_tiled, Bd, ldb, k_load, k_out, b_tiled, P, chunks, ws, dW, ldo,
           col0, db, None)
    ref = A[:, :n_out].astype(np.float64).T @ B[:, :k_out].astype(np.float64)
    np.testing.assert_allclose(dW[:, col0:col0 + k_out], ref, rtol=1e-5, atol=1e-4)   # fp32 sums over up to 300 samples
    untouched = np.ones(ldo, bool)
    untouched[col0:col0 + k_out] = False
    assert np.all(np.isnan(dW[:, untouched]))            # neighbouring columns are left alone
    np.testing.assert_allclose(db, A[:, :n_out].astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("P,chunks", [(100, 3), (5, 7), (1000, 4)])
def test_vecmat(P, chunks):
    rng = np.random.default_rng(P)
    X = rng.standard_normal((P, 256), dtype=np.float32)
    vec4 = rng.standard_normal((P, 4), dtype=np.float32)
    vec = np.ascontiguousarray(vec4.reshape(-1)[3:])        # column 3 of a [P][4] array: stride 4
    ws = np.full(257 * chunks, np.nan, np.float32)
    dv = np.full(256, np.nan, np.float32)
    dvs = np.full(1, np.nan, np.float32)
    H.call("scnerf_vecmat", tile(X, 256), vec, 4, P, chunks, ws, dv, dvs, None)
    v = vec4[:, 3].astype(np.float64)
    np.testing.assert_allclose(dv, v @ X.astype(np.float64), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(dvs[0], v.sum(), rtol=1e-5, atol=1e-5)


def test_split_products_are_fp32_grade():
    """The bf16-split 256 x 256 GEMM against fp64 on operands with full 24-bit significands and a 2^8 spread of
    magnitudes: its error is that of the exact-fp32 MFMA kernel (same bound, measured side by side), far below
    what a plain bf16 (or 2-term) product would give (2^-9 / 2^-17 relative)."""
    rng = np.random.default_rng(5)
    P, chunks = 200, 2
    A = (rng.standard_normal((P, 256)) * np.exp2(rng.integers(-4, 4, (P, 256)))).astype(np.float32)
    B = (rng.standard_normal((P, 256)) * np.exp2(rng.integers(-4, 4, (P, 256)))).astype(np.float32)
    B[rng.random((P, 256)) < 0.5] = 0.0                                  # activations after ReLU
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    scale = np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)
    errs = {}
    for arith in (0, 1):
        assert H.lib().scnerf_wgrad_arithmetic(arith) == arith
        ws = np.full(H.lib().scnerf_wgrad_workspace_floats(256, 256, chunks), np.nan, np.float32)
        dW = np.full((256, 256), np.nan, np.float32)
        db = np.full(256, np.nan, np.float32)
        H.call("scnerf_wgrad", tile(A, 256), 256, 256, 256, 1, tile(B, 256), 256, 256, 256, 1, P, chunks, ws, dW, 256, 0,
               db, None)
        errs[arith] = float((np.abs(dW - ref) / scale).max())
        np.testing.assert_allclose(db, A.astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    assert H.lib().scnerf_wgrad_arithmetic(-1) == 1                      # query leaves the mode alone
    assert H.lib().scnerf_wgrad_arithmetic(2) == 2                       # (the default again)
    assert errs[1] <= 2.0 * errs[0] + 1e-9 and errs[1] < 1e-6, errs
