"""TEST INFRASTRUCTURE: one rank of a ray-parallel training step (north-star config[3] minus the PRD term):
rays of the rank's shard from the learnable camera model, render (NDC, coarse + fine), MSE on both levels,
backward into the flat gradient buffer, ONE all-reduce weighted by the shard sizes.  Used by the CPU test
(gloo, kernels on the SIMT interpreter) and by the GPU test (two processes sharing the one GPU, gloo on device
tensors: RCCL refuses two ranks on one device)."""
import contextlib
import os

import numpy as np
import torch
import torch.distributed as dist

H, W = 24, 32


def build(device):
    from scnerf_amd import synthetic as synth
    from scnerf_amd.create_nerf import FusedNetworkQuery
    from scnerf_amd.run_nerf_helpers import NeRF, get_embedder

    def make(seed):
        net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        net.load_state_dict(synth.network_params(seed=seed))
        net = net.to(device)
        net.flat_parameters()
        return net
    cam, _ = synth.camera_model(H, W, n_cams=5, seed=4, grid_size=4, focal=30.0)
    query = FusedNetworkQuery(get_embedder(10, 0)[0], get_embedder(4, 0)[0])
    return make(0), make(1), cam.to(device), query


def batch(n, sc, sf, device):
    from scnerf_amd import synthetic as synth
    g = torch.Generator().manual_seed(17)
    kps = torch.stack([torch.randint(0, W, (n,), generator=g), torch.randint(0, H, (n,), generator=g)], -1)
    idx = torch.randint(0, 5, (n,), generator=g)
    rnd = synth.render_randoms(n, sc, sf, seed=3)
    return kps.to(device), idx.to(device), synth.target_rgb(n, seed=2).to(device), {k: v.to(device) for k, v in rnd.items()}


def step(nets, lo, hi, n, sc, sf, device, reducer, world_total=None):
    """renders rays [lo, hi) of the global batch and leaves the (reduced) flat gradient in the reducer"""
    from scnerf_amd.get_rays import get_rays_kps_use_camera
    from scnerf_amd.render import render
    net_c, net_f, cam, query = nets
    kps, idx, target, rnd = batch(n, sc, sf, device)
    reducer.zero()
    rays_o, rays_d = get_rays_kps_use_camera(H=H, W=W, camera_model=cam, idx_in_camera_param=idx[lo:hi], kps_list=kps[lo:hi])
    rgb, disp, acc, extras = render(H=H, W=W, chunk=1 << 15, rays=torch.stack([rays_o, rays_d]), retraw=True,
                                    camera_model=cam, mode="train", network_fn=net_c, network_fine=net_f,
                                    network_query_fn=query, N_samples=sc, N_importance=sf, perturb=1.0,
                                    raw_noise_std=1.0, use_viewdirs=True, white_bkgd=False, near=0., far=1.,
                                    _randoms={k: v[lo:hi] for k, v in rnd.items()})
    loss = torch.mean((rgb - target[lo:hi]) ** 2) + torch.mean((extras["rgb0"] - target[lo:hi]) ** 2)
    loss.backward()
    return reducer.all_reduce(hi - lo, world_total if world_total is not None else hi - lo), float(loss.detach())


def worker(rank, world, port, out_dir, device, n, sc, sf, use_emu, with_optimizer):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scnerf_amd.optim import FusedAdam
    from scnerf_amd.parallel import FlatGradAllReduce, shard_rays
    if use_emu:
        from tests.emu.host_on_emu import emulated_device
        ctx = emulated_device()
    else:
        ctx = contextlib.nullcontext()
    dev = torch.device(device)
    with ctx:
        nets = build(dev)
        modules = [nets[0], nets[1], nets[2]]
        if with_optimizer:
            opt = FusedAdam([p for m in modules for p in m.parameters()], lr=5e-4)
            red = FlatGradAllReduce.for_optimizer(opt, world)
        else:
            red = FlatGradAllReduce(modules, world)
        lo, hi = shard_rays(n, rank, world)
        flat, loss = step(nets, lo, hi, n, sc, sf, dev, red, world_total=n)
        np.save(os.path.join(out_dir, "flat%d.npy" % rank), flat.detach().cpu().numpy())
        if with_optimizer:
            opt.grad_sync = None                   # (already reduced above; step() must not reduce again)
            opt.step()
            np.save(os.path.join(out_dir, "param%d.npy" % rank),
                    torch.cat([p.detach().reshape(-1) for m in modules for p in m.parameters()]).cpu().numpy())
        if rank == 0:                              # the same global batch in one process
            nets1 = build(dev)
            if with_optimizer:
                opt1 = FusedAdam([p for m in nets1[:3] for p in m.parameters()], lr=5e-4)
                red1 = FlatGradAllReduce.for_optimizer(opt1, 1)
            else:
                red1 = FlatGradAllReduce(list(nets1[:3]), 1)
            full, _ = step(nets1, 0, n, n, sc, sf, dev, red1)
            np.save(os.path.join(out_dir, "full.npy"), full.detach().cpu().numpy())
    dist.destroy_process_group()
