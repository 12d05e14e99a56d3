"""(GPU box) The fine stage of render_rays as one launch (scnerf_fine_stage_fwd_h3) against its three launches
(fine_sample, mlp_fwd resident, composite_fwd) at 4096 rays x (64 + 128), training mode: ms per call of each.
    python tools/bench_fine_stage.py [--rays 4096] [--sf 128] [--iters 20]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scnerf_amd import mlp_layout as ML, ops, synthetic as synth  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
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
    ap.add_argument("--sf", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    n, sc, sf = a.rays, 64, a.sf
    tot = sc + sf
    p = synth.network_params(seed=1)
    flat = torch.cat([p[name].reshape(-1) for name, _ in ML.PARAM_SHAPES]).cuda()
    wf, rw = ops.pack_weights(flat, "fwd"), ops.pack_resident(flat)
    rays = synth.ray_batch(n, seed=1).cuda()
    z_c = torch.sort(torch.rand(n, sc, device="cuda"), -1)[0]
    w_c = torch.rand(n, sc, device="cuda") ** 4
    u = torch.rand(n, sf, device="cuda")
    noise = torch.randn(n, tot, device="cuda")
    save = ops.save_workspace(n * tot, "cuda")
    mx = ops.ChunkMaxima(n * tot, "cuda")
    z_f, pts_f, _, _, _, _ = ops.fine_sample(rays, z_c, w_c, u)
    raw = ops.mlp_fwd(pts_f, rays[:, 8:11], tot, wf, save, planes=rw, maxima=mx).view(n, tot, 4)
    out = {"rays": n, "samples_per_ray": tot}
    out["fine_sample_ms"] = timed(lambda: ops.fine_sample(rays, z_c, w_c, u), a.iters)
    out["mlp_fwd_train_ms"] = timed(lambda: ops.mlp_fwd(pts_f, rays[:, 8:11], tot, wf, save, planes=rw, maxima=mx), a.iters)
    out["composite_fwd_ms"] = timed(lambda: ops.composite_fwd(raw, z_f, rays, noise, False, want_weights=False), a.iters)
    out["three_launches_ms"] = timed(lambda: (ops.fine_sample(rays, z_c, w_c, u),
                                              ops.mlp_fwd(pts_f, rays[:, 8:11], tot, wf, save, planes=rw, maxima=mx),
                                              ops.composite_fwd(raw, z_f, rays, noise, False, want_weights=False)), a.iters)
    out["fused_fine_stage_ms"] = timed(lambda: ops.fine_stage_fwd(rays, z_c, w_c, u, wf, save, noise, False, rw, maxima=mx), a.iters)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
