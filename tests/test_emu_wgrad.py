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
def test_wgrad(P, lda, n_load, n_out, a_tiled, ldb, k_load, k_out, b_tiled, chunks, ldo, col0):
    """(scnerf_wgrad takes no chunk maxima: the exact-fp32 MFMA kernels in either arithmetic mode)"""
    _wgrad_case(P, lda, n_load, n_out, a_tiled, ldb, k_load, k_out, b_tiled, chunks, ldo, col0)


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


def test_arithmetic_switch():
    lib = H.lib()
    assert lib.scnerf_wgrad_arithmetic(-1) == 1                          # the default: three fp16 products where maxima exist
    assert lib.scnerf_wgrad_arithmetic(0) == 0 and lib.scnerf_wgrad_arithmetic(-1) == 0
    assert lib.scnerf_wgrad_arithmetic(7) == 0                           # anything else only queries
    assert lib.scnerf_wgrad_arithmetic(1) == 1
