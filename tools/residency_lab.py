"""LAB driver for tools/ubench/residency_lab.hip (+ residency_lds_lab.hip): a chain of eight 256 -> 256 ReLU layers on three
fp16 products per product in the candidate residencies of DESIGN.md section 8.  numpy + ctypes only (no torch import on the
GPU box).

    python tools/residency_lab.py --emu --samples 256               # CPU SIMT interpreter: packing + data flow check
    python tools/residency_lab.py --samples 786432 --reps 5         # GPU: check against fp64, then time every kind
    python tools/residency_lab.py --kinds h3p --samples 786432 --reps 3 --no-check      # (what the --pmc passes run)

Prints one JSON line per kind: ms per launch (best / median), ms per layer, issued TFLOP/s, the check's error."""
import argparse
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYERS, W, SECTIONS = 8, 256, 9
KINDS = {"h3": 0, "h3p": 1, "h3-infer": 2, "h3p-infer": 3, "lds": 4, "lds-infer": 5, "h3-compact": 6, "h3-st-default": 7, "h3-st-nt": 8, "h3-st-sc0sc1nt": 9, "h3-st-sc1": 10, "h3-st-sc0sc1": 11, "h3-st-sc1nt": 12, "h3-st-sc0nt": 13}


def lab_input(p, f):
    p = np.asarray(p, np.uint32)[:, None]
    f = np.asarray(f, np.uint32)[None, :]
    with np.errstate(over="ignore"):
        x = p * np.uint32(1103515245) + f * np.uint32(12345) + p * f * np.uint32(7) + np.uint32(0x9e3779b9)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x2c1b3c6d)
        x ^= x >> np.uint32(12)
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def scale_for(bound):
    e = (np.float32(bound).view(np.uint32) >> 23) & 0xff
    return np.uint32((266 - max(13, int(e))) << 23).view(np.float32)


def make_network(seed=0):
    rng = np.random.default_rng(seed)
    a = np.sqrt(6.0 / (2 * W))
    wts = rng.uniform(-a, a, (LAYERS, W, W)).astype(np.float32)
    bias = rng.uniform(-0.05, 0.05, (LAYERS, W)).astype(np.float32)
    sc = np.zeros((SECTIONS, 8), np.float32)
    for l in range(LAYERS):
        sw = scale_for(np.abs(wts[l]).max())
        sc[l, 0], sc[l, 1] = sw, 1.0 / sw
        sc[l, 2] = np.abs(wts[l]).sum(1).max() * (1 + 2.0 ** -20)
        sc[l, 3] = np.abs(bias[l]).max()
    sc[LAYERS, 0] = sc[LAYERS, 1] = 1.0
    return wts, bias, sc


def feature_of_32(sl, g, e):      # mlp_layout.h3_feature_of
    return 32 * (sl >> 1) + 16 * (sl & 1) + 8 * (e >> 2) + 4 * g + (e & 3)


def feature_of_16(sl, g, e):      # mlp_h3p.h: K slab of 32, four lane groups
    return 32 * sl + 16 * (e >> 2) + 4 * g + (e & 3)


def fragment_index(kind_rows):
    """[frags, 64 lanes, 8] (row, col) index pairs of one layer's stream: pairs of output tiles, per pair every K slab,
    per slab [Wh T0][Wh T1][Wl T0][Wl T1]"""
    lane = np.arange(64)
    e = np.arange(8)
    if kind_rows == 32:
        n, g = lane & 31, lane >> 5
        tiles, slabs, fo = 8, 16, feature_of_32
    else:
        n, g = lane & 15, lane >> 4
        tiles, slabs, fo = 16, 8, feature_of_16
    rows, cols, planes = [], [], []
    for P in range(tiles // 2):
        for sl in range(slabs):
            for plane in (0, 1):
                for T in (2 * P, 2 * P + 1):
                    rows.append(np.broadcast_to((kind_rows * T + n)[:, None], (64, 8)))
                    cols.append(fo(sl, g[:, None], e[None, :]))
                    planes.append(plane)
    return np.stack(rows), np.stack(cols), np.array(planes)


def pack_stream(wts, sc, kind_rows):
    rows, cols, planes = fragment_index(kind_rows)
    out = []
    for l in range(LAYERS):
        ws = wts[l].astype(np.float32) * sc[l, 0]                 # exact: a power of two
        hi = ws.astype(np.float16)
        lo = (ws - hi.astype(np.float32)).astype(np.float16)
        both = np.stack([hi, lo])
        out.append(both[planes[:, None, None], rows, cols])
    stream = np.concatenate(out).reshape(-1)
    pad = np.zeros(2 * 32 * 512, np.float16)                      # two chunks the loader runs ahead into
    return np.ascontiguousarray(np.concatenate([stream, pad]))


def pack_stream_lds(wts, sc):
    """candidate (b): per layer and K slab one 16 KB ring slot: [feature tile 8][plane 2] fragments (rows 32 T + n, today's k order)"""
    lane = np.arange(64)
    n, g, e = lane & 31, lane >> 5, np.arange(8)
    out = []
    for l in range(LAYERS):
        ws = wts[l].astype(np.float32) * sc[l, 0]
        hi = ws.astype(np.float16)
        lo = (ws - hi.astype(np.float32)).astype(np.float16)
        for sl in range(16):
            cols = feature_of_32(sl, g[:, None], e[None, :])
            for T in range(8):
                rows = np.broadcast_to((32 * T + n)[:, None], (64, 8))
                out.append(hi[rows, cols])
                out.append(lo[rows, cols])
    stream = np.stack(out).reshape(-1)
    return np.ascontiguousarray(np.concatenate([stream, np.zeros(2 * 32 * 512, np.float16)]))


def check_lds(acc, wts, sc, P):
    """the last layer's accumulators [workgroup][wave][ft][st][r 16][lane 64] against W_7 x0 in fp64"""
    a = acc.reshape(-1, 8, 2, 2, 16, 64)
    wg = np.array([0, a.shape[0] - 1])
    lane = np.arange(64)
    m, g = lane & 31, lane >> 5
    r = np.arange(16)
    worst, sq, cnt = 0.0, 0.0, 0
    s0 = float(scale_for(1.0))
    for w in wg:
        for wave in range(8):
            fb, sb = wave >> 1, wave & 1
            for ft in range(2):
                for st in range(2):
                    feat = 64 * fb + 32 * ft + (r[:, None] & 3) + 8 * (r[:, None] >> 2) + 4 * g[None, :]        # [16, 64]
                    samp = w * 128 + 64 * sb + 32 * st + m                                                  # [64]
                    x0 = lab_input(samp, np.arange(W)).astype(np.float64)                                   # [64, 256]
                    ref = np.einsum("rlk,lk->rl", wts[LAYERS - 1].astype(np.float64)[feat], x0)
                    mag = np.einsum("rlk,lk->rl", np.abs(wts[LAYERS - 1].astype(np.float64))[feat], np.abs(x0))
                    got = a[w, wave, ft, st].astype(np.float64) / (float(sc[LAYERS - 1, 0]) * s0)
                    err = np.abs(got - ref) / mag
                    worst = max(worst, float(err.max()))
                    sq += float((err ** 2).sum())
                    cnt += err.size
    return worst, (sq / cnt) ** 0.5, cnt


def untile(block, P):
    """tile-native section [P/32][t 8][q 4][lane 64][4] -> [P, 256]"""
    b = block.reshape(-1, 8, 4, 2, 32, 4)          # tile, t, q, h, m, j
    return b.transpose(0, 4, 1, 2, 3, 5).reshape(-1, 256)[:P]


def uncompact(block, words, P):
    """kind "h3-compact": a section whose wave-tile blocks hold only the non-zero values, packed in ballot order, + the gate
    words [tiles][lane 64][4] (word 64 r + lane = half (w & 1) of ballot w >> 1; ballot b = ((4 T + q) 4 + e)) -> dense
    [P, 256] and the packed bytes per tile"""
    tiles = (P + 31) // 32
    blk = block.reshape(-1, 32 * 256)[:tiles]
    wd = words.reshape(-1, 64, 4)[:tiles]
    flat = wd.transpose(0, 2, 1).reshape(tiles, 256)                   # word index 64 r + lane
    masks = flat[:, 0::2].astype(np.uint64) | (flat[:, 1::2].astype(np.uint64) << np.uint64(32))      # [tiles, 128]
    lane = np.arange(64, dtype=np.uint64)
    bits = ((masks[:, :, None] >> lane[None, None, :]) & np.uint64(1)).astype(np.int64)                # [tiles, 128, 64]
    order = np.cumsum(bits.reshape(tiles, -1), 1).reshape(tiles, 128, 64) - bits                      # slot of (ballot, lane)
    vals = np.take_along_axis(blk, np.minimum(order.reshape(tiles, -1), 32 * 256 - 1), 1).reshape(tiles, 128, 64) * bits
    # ballot b = (4 T + q) 4 + e, lane = m + 32 h -> feature 32 T + 8 q + 4 h + e of sample m
    v = vals.reshape(tiles, 8, 4, 4, 2, 32)                            # T, q, e, h, m
    dense = v.transpose(0, 5, 1, 2, 4, 3).reshape(tiles * 32, 256)[:P]
    return dense.astype(np.float32), bits.reshape(tiles, -1).sum(1) * 4


def reference(wts, bias, samples):
    x = lab_input(samples, np.arange(W)).astype(np.float64)
    mags = []
    for l in range(LAYERS):
        z = x @ wts[l].astype(np.float64).T + bias[l].astype(np.float64)
        mags.append(np.abs(x) @ np.abs(wts[l].astype(np.float64)).T + np.abs(bias[l]))
        x = np.maximum(z, 0.0)
    return x, mags[-1]


class Hip:
    def __init__(self):
        self.rt = ctypes.CDLL("libamdhip64.so")

    def malloc(self, nbytes):
        p = ctypes.c_void_p()
        assert self.rt.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)) == 0, "hipMalloc(%d)" % nbytes
        return p

    def upload(self, a):
        p = self.malloc(a.nbytes)
        assert self.rt.hipMemcpy(p, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.nbytes), 1) == 0
        return p

    def download(self, p, offset_bytes, out):
        src = ctypes.c_void_p(p.value + offset_bytes)
        assert self.rt.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), src, ctypes.c_size_t(out.nbytes), 2) == 0
        return out

    def sync(self):
        assert self.rt.hipDeviceSynchronize() == 0


def build_emu():
    out_dir = os.path.join(ROOT, "tools", "ubench", "_emu")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libresidency_lab_emu.so")
    shim = os.path.join(ROOT, "tests", "emu", "shim")
    csrc = os.path.join(ROOT, "scnerf_amd", "csrc")
    srcs = [os.path.join(ROOT, "tools", "ubench", f) for f in ("residency_lab.hip", "residency_lds_lab.hip")]
    srcs = [s for s in srcs if os.path.isfile(s)] + [os.path.join(shim, "simt_emu.cpp")]
    deps = srcs + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")] + [os.path.join(shim, "scn_wave.h"),
                                                                                         os.path.join(ROOT, "tools", "ubench", "mlp_h3p.h")]
    if os.path.isfile(out) and all(os.path.getmtime(out) > os.path.getmtime(d) for d in deps):
        return out
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    flags = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-DSCNERF_SIMT_EMU_BUILD=1",
             "-I", shim, "-I", os.path.join(ROOT, "tools", "ubench"), "-I", csrc, "-I", os.path.join(ROOT, "include"),
             "-idirafter", os.path.join(csrc, "device"), "-Wno-psabi",
             "-Wno-unknown-pragmas", "-Wno-unused-variable"]
    objs = []
    for s in srcs:
        o = os.path.join(out_dir, os.path.basename(s).rsplit(".", 1)[0] + ".o")
        subprocess.run([cxx] + flags + ["-x", "c++", "-c", s, "-o", o], check=True)
        objs.append(o)
    subprocess.run([cxx, "-shared", "-o", out] + objs + ["-lpthread"], check=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--samples", type=int, default=786432)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--kinds", default="h3,h3p,h3-infer,h3p-infer")
    ap.add_argument("--lib", default=None)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--zero", action="store_true", help="all-zero weights and biases: what the matrix pipe draws depends on the operand bits")
    ap.add_argument("--loop-seconds", type=float, default=0.0, help="keep launching for this long (power / clock sampling from outside)")
    args = ap.parse_args()
    P = args.samples
    Ppad = (P + 127) // 128 * 128
    wts, bias, sc = make_network()
    if args.zero:
        wts, bias = wts * 0, bias * 0
        args.no_check = True
    lib_path = args.lib or (build_emu() if args.emu else os.path.join(ROOT, "tools", "ubench", "libresidency_lab.so"))
    lib = ctypes.CDLL(lib_path)
    run = lib.residency_lab_run
    run.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
    save_floats = SECTIONS * W * Ppad + SECTIONS * 8 * Ppad
    streams = {32: pack_stream(wts, sc, 32), 16: pack_stream(wts, sc, 16)}
    if any(k.startswith("lds") for k in args.kinds.split(",")):
        streams["lds"] = pack_stream_lds(wts, sc)
        run_lds = lib.residency_lds_lab_run
        run_lds.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
    check_rows = np.unique(np.concatenate([np.arange(0, min(P, 64)), np.arange(max(0, P - 64), P),
                                          (np.arange(16) * 7919 * 32) % max(P - 32, 1)]))
    if os.environ.get("LAB_CHECK_ROWS"):          # (a denser check: every k-th row -- the sampled one misses a 1e-3 corruption rate)
        check_rows = np.unique(np.concatenate([check_rows, np.arange(0, P, int(os.environ["LAB_CHECK_ROWS"]))]))
    ref, mag = (None, None) if args.no_check else reference(wts, bias, check_rows)
    hip = None if args.emu else Hip()
    if hip:
        d_bias, d_sc = hip.upload(bias.reshape(-1)), hip.upload(sc.reshape(-1))
        d_save = hip.malloc(save_floats * 4)
        d_out = hip.malloc(Ppad * 4 * 4)
        d_streams = {k: hip.upload(v) for k, v in streams.items()}
    else:
        save = np.zeros(save_floats, np.float32)
        out = np.zeros(Ppad * 4, np.uint32)
    for name in args.kinds.split(","):
        kind = KINDS[name]
        rows = 32 if name.startswith("h3") and not name.startswith("h3p") else 16
        ms = np.zeros(max(args.reps, 1), np.float32)
        if name.startswith("lds"):
            n_acc = Ppad // 128 * 8 * 4 * 16 * 64
            assert n_acc <= save_floats
            if hip:
                st = run_lds(d_streams["lds"], d_save, P, args.reps, ms.ctypes.data_as(ctypes.c_void_p))
                hip.sync()
                acc = np.zeros(n_acc, np.float32)
                if not args.no_check:
                    hip.download(d_save, 0, acc)
            else:
                st = run_lds(streams["lds"].ctypes.data_as(ctypes.c_void_p), save.ctypes.data_as(ctypes.c_void_p), P, 1,
                             ms.ctypes.data_as(ctypes.c_void_p))
                acc = save[:n_acc]
            assert st == 0, "residency_lds_lab_run -> %d" % st
            rec = {"kind": name, "samples": P, "status": st, "note": "bare: MFMAs + fragment reads + ring + one barrier per K slab"}
            if not args.no_check:
                mx, rms, cnt = check_lds(acc, wts, sc, P)
                rec.update(check_values=cnt, max_err_over_sum_abs=mx, rms_err_over_sum_abs=rms, ok=bool(mx < 2e-6))
            if hip:
                t = np.sort(ms[1:] if len(ms) > 1 else ms)
                flop = 2.0 * LAYERS * W * W * P
                rec.update(ms_best=float(t[0]), ms_median=float(t[len(t) // 2]), ms_per_layer=float(t[0]) / LAYERS,
                           fp32_tflops=flop / float(t[0]) * 1e-9, issued_fp16_tflops=3 * flop / float(t[0]) * 1e-9,
                           frac_of_2500=3 * flop / float(t[0]) * 1e-9 / 2500.0)
            print(json.dumps(rec), flush=True)
            if rec.get("ok") is False:
                sys.exit(1)
            continue
        if hip:
            assert hip.rt.hipMemset(d_save, 0, ctypes.c_size_t(SECTIONS * W * Ppad * 4)) == 0
            st = run(kind, d_streams[rows], d_bias, d_sc, d_save, d_out, P, args.reps, ms.ctypes.data_as(ctypes.c_void_p))
            hip.sync()
        else:
            save[:] = 0
            st = run(kind, streams[rows].ctypes.data_as(ctypes.c_void_p), bias.ctypes.data_as(ctypes.c_void_p),
                     sc.ctypes.data_as(ctypes.c_void_p), save.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
                     P, 1, ms.ctypes.data_as(ctypes.c_void_p))
        assert st == 0, "residency_lab_run(%s) -> %d" % (name, st)
        if hip and args.loop_seconds > 0:
            import time
            t_end = time.time() + args.loop_seconds
            big = np.zeros(200, np.float32)
            while time.time() < t_end:
                run(kind, d_streams[rows], d_bias, d_sc, d_save, d_out, P, 200, big.ctypes.data_as(ctypes.c_void_p))
            ms = big
        rec = {"kind": name, "samples": P, "status": st}
        if not args.no_check and "infer" not in name:
            sect = np.zeros(W * Ppad, np.float32)
            if hip:
                hip.download(d_save, (LAYERS - 1) * W * Ppad * 4, sect)
            else:
                sect[:] = save[(LAYERS - 1) * W * Ppad:LAYERS * W * Ppad]
            if name == "h3-compact":
                words = np.zeros(8 * Ppad, np.uint32)
                if hip:
                    hip.download(d_save, (SECTIONS * W * Ppad + (LAYERS - 1) * 8 * Ppad) * 4, words)
                else:
                    words[:] = save[SECTIONS * W * Ppad + (LAYERS - 1) * 8 * Ppad:][:8 * Ppad].view(np.uint32)
                dense, packed_bytes = uncompact(sect, words, P)
                got = dense[check_rows].astype(np.float64)
                rec["packed_bytes_over_dense_last_layer"] = float(packed_bytes.sum()) / (32.0 * 256 * 4 * len(packed_bytes))
            else:
                got = untile(sect, P)[check_rows].astype(np.float64)
            err = np.abs(got - ref) / mag
            rec["check_rows"] = int(len(check_rows))
            rec["max_err_over_sum_abs"] = float(err.max())
            rec["rms_err_over_sum_abs"] = float(np.sqrt((err ** 2).mean()))
            rec["nonzero_outputs"] = float((got != 0).mean())
            rec["ok"] = bool(err.max() < 2e-6)
        if hip:
            t = np.sort(ms[1:] if len(ms) > 1 else ms)
            best, med = float(t[0]), float(t[len(t) // 2])
            flop = 2.0 * LAYERS * W * W * P
            rec.update(ms_best=best, ms_median=med, ms_per_layer=best / LAYERS, fp32_tflops=flop / best * 1e-9,
                       issued_fp16_tflops=3 * flop / best * 1e-9, frac_of_2500=3 * flop / best * 1e-9 / 2500.0)
        print(json.dumps(rec), flush=True)
        if rec.get("ok") is False:
            sys.exit(1)


if __name__ == "__main__":
    main()
