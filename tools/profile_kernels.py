"""Small fixed workload for counter collection: each heavy kernel a few times at the headline size.
    rocprofv3 --kernel-trace --pmc <counters> -d out -o pmc -- python tools/profile_kernels.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scnerf_amd import mlp_layout as ML, ops, synthetic as synth  # noqa: E402


def main():
    P = 4096 * 192
    p = synth.network_params(seed=0)
    flat = torch.cat([p[n].reshape(-1) for n, _ in ML.PARAM_SHAPES]).cuda()
    wf, wb = ops.pack_weights(flat, "fwd"), ops.pack_weights(flat, "bwd")
    pts = torch.rand(P, 3, device="cuda") * 2 - 1
    vd = torch.nn.functional.normalize(torch.randn(4096, 3, device="cuda"), dim=-1)
    save = ops.save_workspace(P, "cuda")
    d_raw = torch.randn(P, 4, device="cuda")
    reps = int(os.environ.get("REPS", "3"))
    for _ in range(reps):
        ops.mlp_fwd(pts, vd, 192, wf, None)
    for _ in range(reps):
        ops.mlp_fwd(pts, vd, 192, wf, save)
    for _ in range(reps):
        grads, d_pts, d_views = ops.mlp_bwd(d_raw, pts, vd, 192, wb, save)
    for _ in range(reps):
        ops.nerf_wgrad(save, grads, d_raw, P)
    # the same passes with the 256-wide layers as split-arithmetic GEMMs (the default of the training step)
    planes = ops.pack_planes(flat)
    for _ in range(reps):
        ops.mlp_fwd(pts, vd, 192, wf, save, planes=planes)
    for _ in range(reps):
        grads, d_pts, d_views = ops.mlp_bwd(d_raw, pts, vd, 192, wb, save, planes=planes)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
