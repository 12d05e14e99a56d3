"""LAB (GPU): the camera ray generator's backward, many times on the same inputs -- float atomics make the order of its sums
run-to-run different at the 1e-7 level; anything beyond that is a race."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scnerf_amd import synthetic as synth
from scnerf_amd.get_rays import get_rays_kps_use_camera

def main():
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    H, W, n = 24, 32, int(sys.argv[2]) if len(sys.argv) > 2 else 1025
    cam, _ = synth.camera_model(H, W, n_cams=5, seed=4, grid_size=4, focal=30.0)
    cam = cam.cuda()
    g = torch.Generator().manual_seed(17)
    kps = torch.stack([torch.randint(0, W, (n,), generator=g), torch.randint(0, H, (n,), generator=g)], -1).cuda()
    idx = torch.randint(0, 5, (n,), generator=g).cuda()
    go = torch.randn(n, 3, generator=g).cuda() * 1e-3
    gd = torch.randn(n, 3, generator=g).cuda() * 1e-3
    names = ["intrinsics_noise", "extrinsics_noise", "ray_o_noise", "ray_d_noise"]
    ref, worst, bad = None, {k: 0.0 for k in names}, 0
    for it in range(n_iter):
        for k in names:
            getattr(cam, k).grad = None
        ro, rd = get_rays_kps_use_camera(H=H, W=W, camera_model=cam, idx_in_camera_param=idx, kps_list=kps)
        torch.autograd.backward([ro, rd], [go, gd])
        cur = {k: getattr(cam, k).grad.clone() for k in names}
        if ref is None:
            ref = cur
            continue
        for k in names:
            e = float((cur[k] - ref[k]).abs().max() / ref[k].abs().max())
            worst[k] = max(worst[k], e)
            if e > 1e-5:
                bad += 1
                print("iter %d: %s deviates by %.3e of its largest entry" % (it, k, e), flush=True)
    print("iterations %d, outliers %d, worst relative deviation per tensor: %s" % (n_iter, bad, worst))

if __name__ == "__main__":
    main()
