"""TEST / BASELINE INFRASTRUCTURE ONLY -- never imported by the product package.

Where the unmodified reference tree can be found by the checkers (tests/conftest.py, bench.py's cpu_baseline leg):

1. $SCNERF_REFERENCE_ROOT, if set (a maintainer's own checkout);
2. /root/reference (the build container).

The reference does NOT travel to the GPU box in any form (it is Python: no source, no bytecode, no archive): there the
checkers run on the committed golden vectors (tests/golden/, made here by oracle/gen_golden.py from the imported reference)
and the CPU baseline is the pinned oracle (kind "port").  Earlier rounds shipped a git-ignored archive beside the snapshot;
that path is gone.

`ensure()` returns the root or None and exports SCNERF_REFERENCE_ROOT so that oracle/ref_import.py and
tests/dropin_support.py, which read the variable at import, see the same tree.
"""
import os


def _is_tree(root):
    return bool(root) and os.path.isfile(os.path.join(root, "NeRF", "render.py"))


def ensure():
    """-> path of the reference tree, or None when there is none on this machine"""
    env = os.environ.get("SCNERF_REFERENCE_ROOT")
    if env:
        return env if _is_tree(env) else None
    if _is_tree("/root/reference"):
        os.environ["SCNERF_REFERENCE_ROOT"] = "/root/reference"
        return "/root/reference"
    return None
