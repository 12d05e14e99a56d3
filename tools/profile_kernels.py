"""Small fixed workload for counter collection: each heavy kernel of the default step a few times at the headline
size, plus a streaming copy of known size (calibration of FETCH_SIZE / WRITE_SIZE).
    rocprofv3 --kernel-trace --pmc <counters> -d out -o pmc -- python tools/profile_kernels.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scnerf_amd import mlp_layout as ML, ops, synthetic as synth  # noqa: E402

CALIBRATION_FLOATS = 1 << 28        # 1 GiB read + 1 GiB written by one copy kernel


def main():
    P = 4096 * 192
    p = synth.network_params(seed=0)
    flat = torch.cat([p[n].reshape(-1) for n, _ in ML.PARAM_SHAPES]).cuda()
    wf, wb = ops.pack_weights(flat, "fwd"), ops.pack_weights(flat, "bwd")
    rw = ops.pack_resident(flat)
    pts = torch.rand(P, 3, device="cuda") * 2 - 1
    vd = torch.nn.functional.normalize(torch.randn(4096, 3, device="cuda"), dim=-1)
    save = ops.save_workspace(P, "cuda")
    d_raw = torch.randn(P, 4, device="cuda") * 1e-3
    reps = int(os.environ.get("REPS", "3"))
    a = torch.rand(CALIBRATION_FLOATS, device="cuda")
    b = torch.empty_like(a)
    for _ in range(reps):
        b.copy_(a)                          # (the runtime's blit kernel, __amd_rocclr_copyBuffer: wide loads and stores)
    for _ in range(reps):
        ops.mlp_fwd(pts, vd, 192, wf, None, planes=rw)
    mx = ops.ChunkMaxima(P, "cuda")
    for _ in range(reps):
        ops.mlp_fwd(pts, vd, 192, wf, save, planes=rw, maxima=mx)
    # the fused fine stage (what the default step launches): sampler + network + compositing of 4096 rays x (64 + 128)
    rays = synth.ray_batch(4096, seed=1).cuda()
    z_c = torch.sort(torch.rand(4096, 64, device="cuda"), -1)[0]
    w_c = torch.rand(4096, 64, device="cuda") ** 4
    u = torch.rand(4096, 128, device="cuda")
    for _ in range(reps):
        ops.fine_stage_fwd(rays, z_c, w_c, u, wf, None, None, False, rw)
    for _ in range(reps):
        ops.fine_stage_fwd(rays, z_c, w_c, u, wf, save, None, False, rw, maxima=mx)
    for _ in range(reps):
        grads, d_pts, d_views = ops.mlp_bwd(d_raw, pts, vd, 192, wb, save, planes=rw, maxima=mx)
    for _ in range(reps):
        ops.nerf_wgrad(save, grads, d_raw, P, maxima=mx)
    if os.environ.get("WITH_FP32"):
        for _ in range(reps):
            ops.mlp_fwd(pts, vd, 192, wf, save)
        for _ in range(reps):
            ops.mlp_bwd(d_raw, pts, vd, 192, wb, save)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
