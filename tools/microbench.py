"""Kernel micro-timings on the GPU box (HIP events on torch's current stream).
    python tools/microbench.py [--samples 786432]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scnerf_amd import ops, synthetic as synth, mlp_layout as ML  # noqa: E402


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=4096 * 192)
    a = ap.parse_args()
    ops.check_layout()
    p = synth.network_params(seed=0)
    flat = torch.cat([p[n].reshape(-1) for n, _ in ML.PARAM_SHAPES]).cuda()
    res = {}
    wf, wb = ops.pack_weights(flat, "fwd"), ops.pack_weights(flat, "bwd")
    P = a.samples
    pts = (torch.rand(P, 3, device="cuda") * 2 - 1)
    vd = torch.nn.functional.normalize(torch.randn(P // 192, 3, device="cuda"), dim=-1)
    save = ops.save_workspace(P, "cuda")
    d_raw = torch.randn(P, 4, device="cuda")
    flop = 2 * 593408 * P
    for name, sv in (("mlp_fwd_infer", None), ("mlp_fwd_train", save)):
        ms = timeit(lambda: ops.mlp_fwd(pts, vd, 192, wf, sv), iters=5)
        res[name] = {"ms": ms, "tflops": flop / ms / 1e9}
    ms = timeit(lambda: ops.mlp_bwd(d_raw, pts, vd, 192, wb, save), iters=5)
    res["mlp_bwd"] = {"ms": ms, "tflops": flop / ms / 1e9}
    grads, _, _ = ops.mlp_bwd(d_raw, pts, vd, 192, wb, save)
    ms = timeit(lambda: ops.nerf_wgrad(save, grads, d_raw, P), iters=5)
    res["wgrad_all"] = {"ms": ms, "tflops": flop / ms / 1e9}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
