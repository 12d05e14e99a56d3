"""LAB (GPU): what varies when the camera backward's sums vary under two processes time-slicing one GPU (tools/flaky_probe3.py found the
call itself deviating on identical inputs).  Each process repeats the parallel worker's step (that is what creates the contention), then
calls scnerf_camera_rays_bwd K times by hand in three ways, each against its own first result:
  plain    (g_o, g_d)                the retained ray gradients
  cloned   (g_o.clone(), g_d.clone()) the same numbers at fresh addresses
  swapped  (g_d, g_o)                the two buffers exchanged: does a deviation follow the BUFFER or the direction-gradient ARITHMETIC?
Deviating results are kept as .npy under gpurun_out/probe4/ for the ray-level pattern (tools/flaky_probe4.py --analyse)."""
import os, sys
import numpy as np
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "gpurun_out", "probe4")


def loop(rank, iters, K):
    from tests.parallel_nerf_worker import build, batch, H, W
    from scnerf_amd.get_rays import get_rays_kps_use_camera
    from scnerf_amd.render import render
    from scnerf_amd import ops
    os.makedirs(OUT, exist_ok=True)
    dev = torch.device("cuda:0")
    net_c, net_f, cam, query = build(dev)
    n, sc, sf = 1025, 64, 128
    lo, hi = (0, 1025) if rank == 0 else (0, 513)
    kps, idx, target, rnd = batch(n, sc, sf, dev)
    ref, kept, events = {}, 0, {"plain": 0, "cloned": 0, "swapped": 0}
    outs = ("intr", "extr", "grid_o", "grid_d")
    for it in range(iters):
        for m in (net_c, net_f, cam):
            for p in m.parameters(): p.grad = None
        rays_o, rays_d = get_rays_kps_use_camera(H=H, W=W, camera_model=cam, idx_in_camera_param=idx[lo:hi], kps_list=kps[lo:hi])
        rays_o.retain_grad(); rays_d.retain_grad()
        rgb, disp, acc, extras = render(H=H, W=W, chunk=1 << 15, rays=torch.stack([rays_o, rays_d]), retraw=True, camera_model=cam, mode="train",
                                        network_fn=net_c, network_fine=net_f, network_query_fn=query, N_samples=sc, N_importance=sf, perturb=1.0,
                                        raw_noise_std=1.0, use_viewdirs=True, white_bkgd=False, near=0., far=1., _randoms={k: v[lo:hi] for k, v in rnd.items()})
        loss = torch.mean((rgb - target[lo:hi]) ** 2) + torch.mean((extras["rgb0"] - target[lo:hi]) ** 2)
        loss.backward()
        node = rays_d.grad_fn
        g_o, g_d = rays_o.grad.contiguous(), rays_d.grad.contiguous()
        if it == 0:
            np.save(os.path.join(OUT, "inputs_rank%d.npy" % rank), {"g_o": g_o.cpu().numpy(), "g_d": g_d.cpu().numpy(), "kps": kps[lo:hi].cpu().numpy(),
                                                                    "idx": idx[lo:hi].cpu().numpy(), "rays_d": rays_d.detach().cpu().numpy()}, allow_pickle=True)
        for k in range(K):
            for way in ("plain", "cloned", "swapped"):
                a, b = (g_o, g_d) if way == "plain" else (g_o.clone(), g_d.clone()) if way == "cloned" else (g_d, g_o)
                d_in, d_ex, d_go, d_gd, _ = ops.camera_rays_bwd(node.cam, node.cam["n"], a, b)
                cur = dict(zip(outs, (t.cpu().numpy() for t in (d_in, d_ex, d_go, d_gd))))
                if way not in ref:
                    ref[way] = cur
                    for o in outs: np.save(os.path.join(OUT, "ref_rank%d_%s_%s.npy" % (rank, way, o)), cur[o])
                    continue
                rel = {o: float(np.abs(cur[o] - ref[way][o]).max() / (np.abs(ref[way][o]).max() + 1e-30)) for o in outs}
                # translation columns of the pose gradient on their own (they come from g_o in the plain call)
                rel["extr_t"] = float(np.abs(cur["extr"][:, 6:] - ref[way]["extr"][:, 6:]).max() / (np.abs(ref[way]["extr"][:, 6:]).max() + 1e-30))
                if max(rel.values()) > 1e-5:
                    events[way] += 1
                    print("rank %d iter %d call %d %s: %s" % (rank, it, k, way, "  ".join("%s %.2e" % kv for kv in rel.items())), flush=True)
                    if kept < 24:
                        for o in outs: np.save(os.path.join(OUT, "dev_rank%d_%d_%s_%s.npy" % (rank, kept, way, o)), cur[o])
                        kept += 1
    print("rank %d done: %d iterations x %d calls x 3 ways; deviating calls %s" % (rank, iters, K, events), flush=True)


def analyse():
    """which grid cells / cameras moved in each kept event, and by how much of one ray's contribution"""
    import glob
    for f in sorted(glob.glob(os.path.join(OUT, "dev_rank*_grid_d.npy"))):
        base = os.path.basename(f)[4:-len("_grid_d.npy")]
        rank, _, way = base.split("_")[0][4:], base.split("_")[1], base.split("_")[2]
        for o in ("grid_d", "grid_o", "extr", "intr"):
            cur = np.load(os.path.join(OUT, "dev_%s_%s.npy" % (base, o)))
            ref = np.load(os.path.join(OUT, "ref_rank%s_%s_%s.npy" % (rank, way, o)))
            d = cur - ref
            nz = np.argwhere(np.abs(d) > 1e-6 * np.abs(ref).max())
            print("%s %-6s: %d of %d entries moved, largest %.3e (reference max %.3e); first %s" % (base, o, len(nz), d.size, np.abs(d).max(), np.abs(ref).max(), nz[:6].tolist()))


if __name__ == "__main__":
    if sys.argv[1] == "--analyse":
        analyse()
    else:
        mp.spawn(loop, args=(int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 8), nprocs=2, join=True)
