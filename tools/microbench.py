"""Kernel micro-timings on the GPU box (HIP events on torch's current stream).
    python tools/microbench.py [--samples 1048576]"""
import argparse
import json
import sys
import os

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
    ap.add_argument("--samples", type=int, default=4096 * 256)
    a = ap.parse_args()
    ops.check_layout()
    p = synth.network_params(seed=0)
    flat = torch.cat([p[n].reshape(-1) for n, _ in ML.PARAM_SHAPES]).cuda()
    res = {}
    res["pack_fwd_ms"] = timeit(lambda: ops.pack_weights(flat, "fwd"))
    wpk = ops.pack_weights(flat, "fwd")
    P = a.samples
    pts = (torch.rand(P, 3, device="cuda") * 2 - 1)
    vd = torch.nn.functional.normalize(torch.randn(P // 64, 3, device="cuda"), dim=-1)
    save = torch.empty(ML.save_floats(P), device="cuda")
    flop = 2 * 593408 * P
    for name, sv in (("mlp_fwd_infer", None), ("mlp_fwd_train", save)):
        ms = timeit(lambda: ops.mlp_fwd(pts, vd, 64, wpk, sv), iters=5)
        res[name + "_ms"] = ms
        res[name + "_tflops"] = flop / ms / 1e9
    n = 4096
    rays = synth.ray_batch(n).cuda()
    t_vals = torch.linspace(0, 1, 64).cuda()
    t_rand = torch.rand(n, 64, device="cuda")
    res["coarse_sample_ms"] = timeit(lambda: ops.coarse_sample(rays, t_vals, t_rand, False))
    z_c, _ = ops.coarse_sample(rays, t_vals, t_rand, False)
    w_c = torch.rand(n, 64, device="cuda")
    u = torch.rand(n, 128, device="cuda")
    res["fine_sample_ms"] = timeit(lambda: ops.fine_sample(rays, z_c, w_c, u))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
