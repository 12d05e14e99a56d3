"""TEST / BASELINE INFRASTRUCTURE ONLY -- never imported by the product package.

Where the unmodified reference tree can be found by the checkers (tests/conftest.py, bench.py's cpu_baseline leg):

1. $SCNERF_REFERENCE_ROOT, if set;
2. /root/reference (the build container);
3. the git-ignored archive `.ref_ship.tgz` beside the repository (`tools/ship_reference.sh pack`; it travels to the GPU
   box with the snapshot, never into the history), unpacked ONCE into `<repo>/.ref_unpacked/` -- a directory of this
   checkout, created with mode 0700 and owned by the current user (a predictable path under the shared temp directory
   could be pre-created by somebody else), with tarfile's `data` filter (no absolute paths, links out of the tree,
   devices or set-id bits).

`ensure()` returns the root or None and exports SCNERF_REFERENCE_ROOT so that oracle/ref_import.py and
tests/dropin_support.py, which read the variable at import, see the same tree.
"""
import os
import shutil
import tarfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARCHIVE = os.path.join(ROOT, ".ref_ship.tgz")
UNPACKED = os.path.join(ROOT, ".ref_unpacked")
_MARK = os.path.join("reference", "NeRF", "run_nerf.py")


def _is_tree(root):
    return bool(root) and os.path.isfile(os.path.join(root, "NeRF", "render.py"))


def _mine(path):
    st = os.stat(path)
    return st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0


def ensure():
    """-> path of the reference tree, or None when there is none on this machine"""
    env = os.environ.get("SCNERF_REFERENCE_ROOT")
    if env:
        return env if _is_tree(env) else None
    if _is_tree("/root/reference"):
        return "/root/reference"
    if not os.path.isfile(ARCHIVE):
        return None
    if not (os.path.isdir(UNPACKED) and _mine(UNPACKED) and os.path.isfile(os.path.join(UNPACKED, _MARK))):
        tmp = "%s.%d" % (UNPACKED, os.getpid())
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp, mode=0o700)
        with tarfile.open(ARCHIVE) as tf:
            tf.extractall(tmp, filter="data")
        try:
            if os.path.isdir(UNPACKED):                 # stale, partial or not ours
                shutil.rmtree(UNPACKED)
            os.rename(tmp, UNPACKED)
        except OSError:                                 # another process of this checkout was first
            shutil.rmtree(tmp, ignore_errors=True)
            if not (_mine(UNPACKED) and os.path.isfile(os.path.join(UNPACKED, _MARK))):
                return None
    root = os.path.join(UNPACKED, "reference")
    os.environ["SCNERF_REFERENCE_ROOT"] = root
    return root
