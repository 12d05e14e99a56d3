"""Every compute entry point of the C ABI accepts an empty batch (n = 0) as a no-op returning 0, and
rejects null outputs / impossible sizes with an argument error instead of launching (kernel sources under
the CPU SIMT interpreter)."""
import ctypes

import numpy as np
import pytest

from scnerf_amd import mlp_layout as ML
from tests.emu import harness as H

pytestmark = pytest.mark.emu
F = ctypes.c_float


def test_core_entry_points_accept_empty_batches():
    z = np.zeros(0, np.float32)
    zl = np.zeros(0, np.int64)
    one = np.zeros(16, np.float32)
    H.call("scnerf_searchsorted", z, z, zl, 0, 1, 1, 5, 4, 0, None)
    H.call("scnerf_sample_pdf", z, z, z, 4, z, zl, z, 0, 9, 4, None)
    H.call("scnerf_coarse_sample", z, 8, one, None, z, z, 0, 8, 0, None)
    H.call("scnerf_fine_sample", z, 8, z, z, z, 4, z, z, z, z, zl, z, 0, 8, 4, None)
    H.call("scnerf_composite_fwd", z, z, z, 8, None, 0, z, z, z, z, z, 0, 8, None)
    H.call("scnerf_composite_bwd", z, z, z, 8, None, 0, None, None, None, None, None, z, z, 0, 8, None)
    H.call("scnerf_ray_reduce", z, z, z, None, z, 11, 0, 0, 8, None)
    H.call("scnerf_gather_f32", one, np.zeros(0, np.int32), z, 0, None)
    for pd in (3, 4):
        wf = np.zeros(ML.layout(pd).fwd_total, np.float32)
        wb = np.zeros(ML.layout(pd).bwd_total, np.float32)
        H.call("scnerf_mlp_fwd", pd, z, one, 3, 1, wf, z, None, 0, None)
        H.call("scnerf_mlp_bwd", pd, z, z, one, 3, 1, wb, one, one, z, z, 0, None)
    H.call("scnerf_adam_step", z, z, z, z, 0, ctypes.c_double(1e-3), ctypes.c_double(0.9), ctypes.c_double(0.999),
           ctypes.c_double(1e-8), ctypes.c_double(0.0), 1, None)
    H.call("scnerf_prd_loss_fwd", z, z, z, z, z, z, one, np.zeros(32, np.float32), F(1e-10), F(5.0), 1, 0, 0,
           np.zeros(6, np.float32), np.zeros(1, np.float32), np.zeros(1, np.float32), None)


@pytest.mark.parametrize("name,args", [
    ("scnerf_sample_pdf", lambda z, zl: (z, z, z, 4, None, zl, z, 1, 9, 4, None)),                  # null output
    ("scnerf_composite_fwd", lambda z, zl: (z, z, z, 3, None, 0, z, z, z, z, z, 1, 8, None)),         # ray_stride < 6
    ("scnerf_mlp_fwd", lambda z, zl: (5, z, z, 3, 1, z, z, None, 1, None)),                           # unknown variant
    ("scnerf_wgrad", lambda z, zl: (z, 6, 6, 6, 0, z, 8, 8, 8, 0, 4, 1, z, z, 8, 0, None, None)),     # lda % 4 != 0
])
def test_argument_errors_do_not_launch(name, args):
    z = np.zeros(64, np.float32)
    zl = np.zeros(64, np.int64)
    with pytest.raises(AssertionError, match="returned -"):
        H.call(name, *args(z, zl))
