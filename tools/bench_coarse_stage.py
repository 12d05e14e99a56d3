"""(GPU box) The coarse stage of render_rays as one launch (scnerf_coarse_stage_fwd) against its three launches
(coarse_sample, mlp_fwd resident, composite_fwd) at 4096 rays x 64, training mode: ms per call of each.
    python tools/bench_coarse_stage.py [--rays 4096] [--iters 20]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scnerf_amd import mlp_layout as ML, ops, synthetic as synth  # noqa: E402
from scnerf_amd.functional import host_linspace  # noqa: E402
from tools.bench_fine_stage import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    n, sc = a.rays, 64
    p = synth.network_params(seed=1)
    flat = torch.cat([p[name].reshape(-1) for name, _ in ML.PARAM_SHAPES]).cuda()
    wf, rw = ops.pack_weights(flat, "fwd"), ops.pack_resident(flat)
    rays = synth.ray_batch(n, seed=1).cuda()
    t_vals = host_linspace(sc, rays.device)
    t_rand = torch.rand(n, sc, device="cuda")
    noise = torch.randn(n, sc, device="cuda")
    save = ops.save_workspace(n * sc, "cuda")
    mx = ops.ChunkMaxima(n * sc, "cuda")
    z, pts = ops.coarse_sample(rays, t_vals, t_rand, False)
    raw = ops.mlp_fwd(pts, rays[:, 8:11], sc, wf, save, planes=rw, maxima=mx).view(n, sc, 4)
    out = {"rays": n, "samples_per_ray": sc}
    out["coarse_sample_ms"] = timed(lambda: ops.coarse_sample(rays, t_vals, t_rand, False), a.iters)
    out["mlp_fwd_train_ms"] = timed(lambda: ops.mlp_fwd(pts, rays[:, 8:11], sc, wf, save, planes=rw, maxima=mx), a.iters)
    out["composite_fwd_ms"] = timed(lambda: ops.composite_fwd(raw, z, rays, noise, False), a.iters)
    out["three_launches_ms"] = timed(lambda: (ops.coarse_sample(rays, t_vals, t_rand, False),
                                              ops.mlp_fwd(pts, rays[:, 8:11], sc, wf, save, planes=rw, maxima=mx),
                                              ops.composite_fwd(raw, z, rays, noise, False)), a.iters)
    out["fused_coarse_stage_ms"] = timed(lambda: ops.coarse_stage_fwd(rays, t_vals, t_rand, False, wf, save, noise, False,
                                                                      planes=rw, maxima=mx), a.iters)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
