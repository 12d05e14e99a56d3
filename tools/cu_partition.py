"""LAB (GPU): the HBM-bound weight-gradient group of the fine pass beside the matrix-pipe-bound data-gradient chain of the
coarse pass, each on its own share of the CUs (hipExtStreamCreateWithCUMask), against the same two calls back to back on
one stream.  Round 4's two-stream run gained nothing because each kernel's workgroups fill the chip in turn; a CU mask
makes the two run side by side.

    python tools/cu_partition.py [--splits 256:0,192:64,176:80,160:96,144:112,128:128] > profiles/r05_cu_partition.txt

A split W:D gives W CUs to the weight-gradient group and D to the data gradients; the weight-gradient chunking follows W
(one big-GEMM workgroup per CU of its share).  Mask bit i is CU i in the driver's enumeration, which deals consecutive bits
round-robin over the eight XCDs, so the lowest W bits are W / 8 CUs of every XCD."""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from scnerf_amd import mlp_layout as ML, ops, synthetic as synth      # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(lo, hi):
    """a stream confined to CUs lo .. hi - 1 (mask bits)"""
    words = (ctypes.c_uint32 * 8)()
    for b in range(lo, hi):
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    st = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert st == 0, "hipExtStreamCreateWithCUMask -> %d" % st
    return torch.cuda.ExternalStream(s.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splits", default="256:0,208:48,192:64,176:80,160:96,144:112,128:128")
    ap.add_argument("--iters", type=int, default=12)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lay = ML.layout(3)
    p = synth.network_params(seed=1)
    flat = torch.cat([p[name].reshape(-1) for name, _ in lay.param_shapes]).float().to(dev)
    wpk, wbk, rw = ops.pack_weights(flat, "fwd"), ops.pack_weights(flat, "bwd"), ops.pack_resident(flat)
    rays = 4096
    g = torch.Generator().manual_seed(5)
    vd = torch.randn(rays, 3, generator=g)
    vd = (vd / vd.norm(dim=-1, keepdim=True)).to(dev)
    flat_grad = torch.zeros(lay.n_params, device=dev)
    base_chunks = ops.wgrad_chunks

    def prepare(spr, chunks):
        """one pass's forward + data gradients, with the weight-gradient chunking `chunks`"""
        P = rays * spr
        ops.wgrad_chunks = (lambda P_: chunks) if chunks else base_chunks
        pts = (torch.rand(P, 3, generator=g) * 3 - 1.5).to(dev)
        d_raw = (torch.randn(P, 4, generator=g) * 1e-3).to(dev)
        save = ops.save_workspace(P, dev)
        mx = ops.ChunkMaxima(P, dev)
        ops.mlp_fwd_resident(pts, vd, spr, wpk, rw, save, maxima=mx)
        grads, _, _ = ops.mlp_bwd_resident(d_raw, pts, vd, spr, wbk, rw, save, maxima=mx)
        return dict(P=P, spr=spr, pts=pts, d_raw=d_raw, save=save, mx=mx, grads=grads, chunks=chunks)

    def wgrad(f):
        ops.wgrad_chunks = (lambda P_: f["chunks"]) if f["chunks"] else base_chunks
        ops.nerf_wgrad(f["save"], f["grads"], f["d_raw"], f["P"], flat_grad=flat_grad, maxima=f["mx"])

    def dgrad(c):
        ops.mlp_bwd_resident(c["d_raw"], c["pts"], vd, c["spr"], wbk, rw, c["save"], maxima=c["mx"])

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    coarse = prepare(64, 0)
    fine = prepare(192, 0)
    t_w, t_d = timed(lambda: wgrad(fine)), timed(lambda: dgrad(coarse))
    t_serial = timed(lambda: (dgrad(coarse), wgrad(fine)))
    print("# weight gradients of the fine pass (P = 786 432) beside the data gradients of the coarse pass (P = 262 144), ms")
    print("# alone, whole chip: weight-gradient group %.3f, data gradients %.3f; back to back on one stream %.3f" % (t_w, t_d, t_serial))
    print("# split W:D   wgrad alone on W   dgrad alone on D   side by side   saved vs back to back")
    main_s = torch.cuda.current_stream()
    rows = []
    for sp in a.splits.split(","):
        W, D = (int(x) for x in sp.split(":"))
        if D == 0:
            continue
        sW, sD = masked_stream(0, W), masked_stream(W, W + D)
        f = prepare(192, W)

        def on(stream, fn):
            def run():
                ev = torch.cuda.Event()
                ev.record(main_s)
                stream.wait_event(ev)
                with torch.cuda.stream(stream):
                    fn()
                done = torch.cuda.Event()
                done.record(stream)
                main_s.wait_event(done)
            return run

        def both():
            ev = torch.cuda.Event()
            ev.record(main_s)
            sW.wait_event(ev)
            sD.wait_event(ev)
            with torch.cuda.stream(sD):
                dgrad(coarse)
            with torch.cuda.stream(sW):
                wgrad(f)
            for s in (sW, sD):
                done = torch.cuda.Event()
                done.record(s)
                main_s.wait_event(done)

        tw, td, tb = timed(on(sW, lambda: wgrad(f))), timed(on(sD, lambda: dgrad(coarse))), timed(both)
        rows.append({"split": sp, "wgrad_alone": tw, "dgrad_alone": td, "side_by_side": tb, "saved": t_serial - tb})
        print("%9s %14.3f %18.3f %14.3f %18.3f" % (sp, tw, td, tb, t_serial - tb), flush=True)
    print(json.dumps({"alone_wgrad": t_w, "alone_dgrad": t_d, "serial": t_serial, "rows": rows}))


if __name__ == "__main__":
    main()
