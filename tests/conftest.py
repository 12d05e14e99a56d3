import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def _locate_reference():
    """TEST INFRASTRUCTURE.  The reference tree exists only in the build container (or where a maintainer points
    SCNERF_REFERENCE_ROOT at a checkout): oracle/ref_ship.py finds it for the tests that import the unmodified reference
    (tests/dropin_support.py, oracle/ref_import.py).  On the GPU box there is none -- nothing of the reference travels --
    and those tests skip; everything else runs on the committed golden vectors."""
    from oracle import ref_ship
    ref_ship.ensure()


_locate_reference()

if torch.cuda.is_available():
    # The GPU suite runs serially and most of its wall time is the CPU oracle (torch-CPU at up to 4096 rays): on the GPU box's
    # 256 hardware threads torch's default intra-op pool oversubscribes the memory system -- 32 threads are the fastest
    # (bench.py's probe of the reference's CPU step finds the same): 150 -> 91 s for the suite.
    torch.set_num_threads(min(32, os.cpu_count() or 1))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "emu: runs the HIP kernel sources under the CPU SIMT interpreter")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def pytest_xdist_auto_num_workers(config):
    """`-n auto` (pytest.ini) means: four workers for the CPU suite (the SIMT interpreter's long tests are serial),
    none on the GPU box -- one GPU, and the two-process RCCL tests want it to themselves."""
    expr = (config.getoption("markexpr", "") or "").strip()
    return 4 if expr == "not gpu" and not torch.cuda.is_available() else 0


def pytest_sessionfinish(session, exitstatus):
    try:
        from tests import parity_attribution
        parity_attribution.write_report()
    except Exception:            # reporting must never turn a green run red
        pass


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))
