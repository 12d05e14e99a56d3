"""Microbenchmark + parity probe of the resident-arithmetic kernels (csrc/mlp_fwd_h3.hip, mlp_bwd_h3.hip) against the
fused fp32 kernels, on the GPU:   python tools/bench_h3.py [--rays 4096] [--out file.json]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from scnerf_amd import mlp_layout as ML, ops, synthetic as synth      # noqa: E402


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--spr", type=int, default=192)
    ap.add_argument("--out", default=None)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only-resident-train", action="store_true", help="time the two resident training kernels only (ablation builds)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    pd = 3
    lay = ML.layout(pd)
    p = synth.network_params(seed=1)
    flat = torch.cat([p[name].reshape(-1) for name, _ in lay.param_shapes]).float().to(dev)
    wpk = ops.pack_weights(flat, "fwd")
    wbk = ops.pack_weights(flat, "bwd")
    rw = ops.pack_resident(flat)
    P = a.rays * a.spr
    g = torch.Generator().manual_seed(5)
    pts = (torch.rand(P, 3, generator=g) * 3 - 1.5).to(dev)
    vd = torch.randn(a.rays, 3, generator=g)
    vd = (vd / vd.norm(dim=-1, keepdim=True)).to(dev)
    save_a = ops.save_workspace(P, dev)
    save_b = ops.save_workspace(P, dev)
    out = {"P": P, "flop_per_pass": 2 * 593408 * P}
    res = {}

    def run(name, fn):
        ms = timed(fn, a.iters)
        res[name] = {"ms": ms, "tflops_fp32_equiv": out["flop_per_pass"] / ms / 1e9}
        print("%-34s %8.3f ms  %7.1f TFLOP/s (algorithmic fp32)" % (name, ms, res[name]["tflops_fp32_equiv"]), flush=True)

    if a.only_resident_train:
        # (ablation builds, tools/ablate_h3_run.sh) time + the clock and socket power each kernel sustains over 1.5 s of
        # back-to-back launches: a variant can be faster in CYCLES or in CLOCK (the socket sits at its power cap)
        import bench
        tel = bench.Telemetry(dev)
        d_raw = torch.randn(P, 4, generator=g).to(dev) * 1e-3
        rec = {}
        for name, fn in (("resident train", lambda: ops.mlp_fwd_resident(pts, vd, a.spr, wpk, rw, save_b)),
                         ("resident dgrad", lambda: ops.mlp_bwd_resident(d_raw, pts, vd, a.spr, wbk, rw, save_b))):
            run(name, fn)
            rec[name] = round(res[name]["ms"], 3)
            t = tel.sample_while(fn, 1.5)
            if t:
                rec[name + " sustained"] = {"ms": round(t["ms_per_step_during"], 3), "ghz": round(t["clock_ghz"], 3),
                                            "watt": round(t["socket_power_w"])}
        print(json.dumps(rec))
        return
    run("fused fp32 train", lambda: ops.mlp_fwd(pts, vd, a.spr, wpk, save_a))
    run("fused fp32 infer", lambda: ops.mlp_fwd(pts, vd, a.spr, wpk, None))
    run("resident train", lambda: ops.mlp_fwd_resident(pts, vd, a.spr, wpk, rw, save_b))
    run("resident infer", lambda: ops.mlp_fwd_resident(pts, vd, a.spr, wpk, rw, None))
    run("pack_resident", lambda: ops.pack_resident(flat, out=rw))

    # ---- data-gradient chain ----
    d_raw = torch.randn(P, 4, generator=g).to(dev) * 1e-3
    ops.mlp_fwd(pts, vd, a.spr, wpk, save_a)
    run("fused fp32 dgrad", lambda: ops.mlp_bwd(d_raw, pts, vd, a.spr, wbk, save_a))
    run("resident dgrad", lambda: ops.mlp_bwd_resident(d_raw, pts, vd, a.spr, wbk, rw, save_a))
    ga, dpa, dva = ops.mlp_bwd(d_raw, pts, vd, a.spr, wbk, save_a)
    gb, dpb, dvb = ops.mlp_bwd_resident(d_raw, pts, vd, a.spr, wbk, rw, save_a)
    torch.cuda.synchronize()
    bpar = {"d_pts": {"max_abs_diff": float((dpa - dpb).abs().max()), "max": float(dpa.abs().max())},
            "d_views": {"max_abs_diff": float((dva - dvb).abs().max()), "max": float(dva.abs().max())}}
    goff, _ = ML.section_offsets(ML.GRAD_SECTIONS, P)
    Ppg = ML.padded_samples(P)
    for name, wd in ML.GRAD_SECTIONS:
        xa = ga[goff[name]: goff[name] + wd * Ppg]
        xb = gb[goff[name]: goff[name] + wd * Ppg]
        bpar[name] = {"max_abs_diff": float((xa - xb).abs().max()), "max": float(xa.abs().max()),
                      "rms_diff": float((xa - xb).pow(2).mean().sqrt()), "rms": float(xa.pow(2).mean().sqrt())}
    out["parity_dgrad_vs_fused_fp32"] = bpar
    print(json.dumps(bpar, indent=1))
    del ga, gb

    # parity: resident vs fused fp32 (both against each other; the oracle comparison lives in tests/)
    raw_a = ops.mlp_fwd(pts, vd, a.spr, wpk, save_a)
    raw_b = ops.mlp_fwd_resident(pts, vd, a.spr, wpk, rw, save_b)
    torch.cuda.synchronize()
    par = {"raw_max_abs": float((raw_a - raw_b).abs().max()), "raw_max": float(raw_a.abs().max())}
    off, total = ML.section_offsets(lay.save_sections, P)
    Pp = ML.padded_samples(P)
    for name, w in lay.save_sections:
        xa = save_a[off[name]: off[name] + w * Pp]
        xb = save_b[off[name]: off[name] + w * Pp]
        par[name] = {"max_abs_diff": float((xa - xb).abs().max()), "max": float(xa.abs().max())}
    ma = save_a[total:].view(torch.int32)
    mb = save_b[total:].view(torch.int32)
    diff_bits = (ma ^ mb)
    par["mask_words_differing"] = int((diff_bits != 0).sum())
    par["mask_words"] = int(ma.numel())
    out["timing"] = res
    out["parity_vs_fused_fp32"] = par
    print(json.dumps(par, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
