"""Builds libscnerf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m scnerf_amd.csrc.build [--force] [--save-temps]

One translation unit per .hip file, linked into scnerf_amd/libscnerf_hip.so (in-tree, so
that it travels to the GPU box with the repository snapshot)."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libscnerf_hip.so")
OBJ = os.path.join(HERE, "_obj")

ARCH = "gfx950"
# -ffp-contract=off: the sampling / compositing / camera arithmetic must round like the
# reference's op-by-op fp32 tensor code (no fused multiply-add where it has mul then add).
# The MLP kernels use explicit MFMA / fmaf and are unaffected.
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
          "-fno-fast-math", "-I", os.path.join(HERE, "device"), "-I", HERE,
          "-I", os.path.join(ROOT, "include"), "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hip"))


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode() + b"\0" + f.read())
    h.update(" ".join(COMMON).encode())
    return h.hexdigest()


def _headers():
    hs = []
    for d in (HERE, os.path.join(HERE, "device"), os.path.join(ROOT, "include")):
        hs += [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")]
    return hs


def build(force=False, save_temps=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    stamp = os.path.join(OBJ, "stamp.txt")
    dig = _digest(srcs + _headers())
    if not force and os.path.isfile(OUT) and os.path.isfile(stamp) and open(stamp).read() == dig:
        return OUT
    hdig = _digest(_headers())

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        tag = obj + ".sha"
        d = _digest([src]) + hdig
        if not force and os.path.isfile(obj) and os.path.isfile(tag) and open(tag).read() == d:
            return obj
        cmd = ["hipcc"] + COMMON + ["-c", src, "-o", obj]
        if save_temps:
            cmd += ["-save-temps=obj"]
        if verbose:
            print("[build]", os.path.basename(src), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)
        open(tag, "w").write(d)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    open(stamp, "w").write(dig)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv))
