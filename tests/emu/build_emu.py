"""TEST INFRASTRUCTURE: compiles the product's .hip sources for the host with the CPU
SIMT interpreter's shim headers first on the include path -> tests/emu/_build/libscnerf_emu.so.
The result exports the same C-ABI as libscnerf_hip.so but takes host pointers; it is loaded
only by tests (tests/emu/harness.py), never by the scnerf_amd package."""
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "scnerf_amd", "csrc")
OUTDIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUTDIR, "libscnerf_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build(verbose=False):
    """(serialised across processes: pytest-xdist workers all ask for the library at start-up)"""
    import fcntl
    os.makedirs(OUTDIR, exist_ok=True)
    with open(os.path.join(OUTDIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build(verbose=False):
    os.makedirs(OUTDIR, exist_ok=True)
    cxx = CLANG if os.path.isfile(CLANG) else "clang++"
    flags = ["-O2", "-g1", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma",
             "-x", "c++", "-DSCNERF_SIMT_EMU_BUILD=1",
             "-I", os.path.join(HERE, "shim"), "-I", CSRC, "-I", os.path.join(ROOT, "include"),
             "-idirafter", os.path.join(CSRC, "device"),        # scn_lab.h (scn_wave.h is the shim's: found first)
             "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-unused-variable"]
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    srcs.append(os.path.join(HERE, "shim", "simt_emu.cpp"))
    hdrs = []
    for d in (CSRC, os.path.join(CSRC, "device"), os.path.join(HERE, "shim"), os.path.join(HERE, "shim", "hip"), os.path.join(ROOT, "include")):
        hdrs += [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")]
    h = hashlib.sha256()
    for p in sorted(hdrs):
        h.update(open(p, "rb").read())
    h.update(" ".join(flags).encode())
    hd = h.hexdigest()

    def one(src):
        obj = os.path.join(OUTDIR, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        tag = obj + ".sha"
        d = hashlib.sha256(open(src, "rb").read()).hexdigest() + hd
        if os.path.isfile(obj) and os.path.isfile(tag) and open(tag).read() == d:
            return obj, False
        if verbose:
            print("[emu build]", os.path.basename(src), flush=True)
        r = subprocess.run([cxx] + flags + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emu compile failed for %s:\n%s" % (src, r.stderr[-6000:]))
        open(tag, "w").write(d)
        return obj, True

    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(one, srcs))
    if any(c for _, c in res) or not os.path.isfile(OUT):
        r = subprocess.run([cxx, "-shared", "-o", OUT] + [o for o, _ in res] + ["-lpthread"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emu link failed:\n" + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(verbose=True))
