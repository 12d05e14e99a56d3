"""LAB (GPU): two processes time-slicing one GPU, each repeating the SAME render_rays training step and comparing the ray
gradients d(loss)/d(ray_batch) bit for bit with its first run -- localises the rare camera-block mismatch of
tools/flaky_probe.py to a column group of the ray gradient (origins 0:3, directions 3:6, view directions 8:11)."""
import os, sys
import numpy as np
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def loop(rank, iters):
    from scnerf_amd import synthetic as synth
    from scnerf_amd.create_nerf import FusedNetworkQuery
    from scnerf_amd.render import render_rays
    from scnerf_amd.run_nerf_helpers import NeRF, get_embedder
    dev = torch.device("cuda:0")
    def make(seed):
        net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        net.load_state_dict(synth.network_params(seed=seed)); net = net.to(dev); net.flat_parameters(); return net
    net_c, net_f = make(0), make(1)
    query = FusedNetworkQuery(get_embedder(10, 0)[0], get_embedder(4, 0)[0])
    n = 1025 if rank == 0 else 513
    rays0 = synth.ray_batch(n, seed=11).to(dev)
    rnd = {k: v.to(dev) for k, v in synth.render_randoms(n, 64, 128, seed=12).items()}
    target = torch.rand(n, 3, generator=torch.Generator().manual_seed(13)).to(dev)
    ref = None
    for it in range(iters):
        rays = rays0.clone().requires_grad_(True)
        for net in (net_c, net_f):
            for p in net.parameters(): p.grad = None
        ret = render_rays(rays, net_c, query, 64, retraw=True, perturb=1.0, N_importance=128, network_fine=net_f, raw_noise_std=1.0, _randoms=rnd)
        loss = torch.mean((ret["rgb_map"] - target) ** 2) + torch.mean((ret["rgb0"] - target) ** 2)
        loss.backward()
        g = rays.grad.detach().cpu().numpy()
        w = net_f.flat_parameters().grad
        wsum = None if w is None else float(w.double().abs().sum())
        if ref is None:
            ref, wref = g, wsum
            continue
        if not np.array_equal(g, ref) or wsum != wref:
            d = np.abs(g - ref)
            rows = np.nonzero(d.max(1) > 0)[0]
            print("rank %d iter %d: ray gradients differ in %d rays; max |diff| by column group: o %.3e  d %.3e  viewdirs %.3e (scale %.3e); weight-gradient checksum equal: %s; rays %s"
                  % (rank, it, len(rows), d[:, 0:3].max(), d[:, 3:6].max(), d[:, 8:11].max(), np.abs(ref).max(), wsum == wref, rows[:12]), flush=True)
    print("rank %d done, %d iterations" % (rank, iters), flush=True)

if __name__ == "__main__":
    mp.spawn(loop, args=(int(sys.argv[1]),), nprocs=2, join=True)
