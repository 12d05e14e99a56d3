"""Weight-gradient GEMM (HIP source under the CPU SIMT interpreter) vs numpy."""
import numpy as np
import pytest

from tests.emu import harness as H

pytestmark = pytest.mark.emu


@pytest.mark.parametrize("P,lda,n_load,n_out,ldb,k_load,k_out,chunks,with_vec,ldo,col0", [
    (100, 256, 256, 256, 256, 256, 256, 3, True, 256, 0),     # trunk layer / feature (+alpha side product)
    (77, 256, 256, 256, 64, 64, 63, 2, False, 319, 0),        # layer 0 / skip part: encoded points
    (50, 256, 256, 256, 256, 256, 256, 1, False, 319, 63),    # skip layer, h part at column offset 63
    (64, 128, 128, 128, 256, 256, 256, 2, False, 283, 0),     # views layer, feature part
    (33, 128, 128, 128, 32, 32, 27, 4, False, 283, 256),      # views layer, encoded direction part
    (90, 4, 4, 3, 128, 128, 128, 2, False, 128, 0),           # rgb layer from d_raw
    (5, 256, 256, 256, 256, 256, 256, 7, True, 256, 0),       # more chunks than stages
])
def test_wgrad(P, lda, n_load, n_out, ldb, k_load, k_out, chunks, with_vec, ldo, col0):
    rng = np.random.default_rng(P + n_load + k_load)
    A = rng.standard_normal((P, lda), dtype=np.float32)
    B = rng.standard_normal((P, ldb), dtype=np.float32)
    vec4 = rng.standard_normal((P, 4), dtype=np.float32)
    ws_n = H.lib().scnerf_wgrad_workspace_floats(n_load, k_load, chunks)
    assert ws_n > 0
    ws = np.full(ws_n, np.nan, np.float32)
    dW = np.full((n_out, ldo), np.nan, np.float32)
    db = np.full(n_out, np.nan, np.float32)
    dv = np.full(k_out, np.nan, np.float32)
    dvs = np.full(1, np.nan, np.float32)
    vec = np.ascontiguousarray(vec4.reshape(-1)[3:]) if with_vec else None
    # (vec points at column 3 of a [P][4] array: stride 4)
    H.call("scnerf_wgrad", A, lda, n_load, n_out, B, ldb, k_load, k_out,
           vec, 4, P, chunks, ws, dW, ldo, col0, db, dv if with_vec else None, dvs if with_vec else None, None)
    ref = A[:, :n_out].astype(np.float64).T @ B[:, :k_out].astype(np.float64)
    np.testing.assert_allclose(dW[:, col0:col0 + k_out], ref, rtol=1e-5, atol=1e-5)
    untouched = np.ones(ldo, bool)
    untouched[col0:col0 + k_out] = False
    assert np.all(np.isnan(dW[:, untouched]))            # neighbouring columns are left alone
    np.testing.assert_allclose(db, A[:, :n_out].astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    if with_vec:
        v = vec4[:, 3].astype(np.float64)
        np.testing.assert_allclose(dv, v @ B[:, :k_out].astype(np.float64), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(dvs[0], v.sum(), rtol=1e-5, atol=1e-5)
