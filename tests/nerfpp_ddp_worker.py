"""TEST INFRASTRUCTURE: one rank of a NeRF++ data-parallel step written like the reference's loop
(nerfplusplus/ddp_train_nerf.py:400-470: zero_grad / forward of every cascade level / backward / step) on models made
by scnerf_amd.nerfplusplus.create_nerf under an initialised process group -- gradient synchronisation must come from
`optim.step()` itself (the reference got it from DistributedDataParallel).  Kernels on the SIMT interpreter."""
import os
import types

import numpy as np
import torch
import torch.distributed as dist


def worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scnerf_amd import synthetic as synth
    from scnerf_amd.nerfplusplus import ddp_train_nerf as TR
    from scnerf_amd.nerfplusplus.create_nerf import create_nerf
    from tests.emu.host_on_emu import emulated_device
    Hh, Ww = 20, 30
    spec = synth.camera_spec(Hh, Ww, n_cams=3, seed=33, multiplicative=True, focal=25.0)
    args = types.SimpleNamespace(
        max_freq_log2=10, max_freq_log2_viewdirs=4, netdepth=8, netwidth=256, use_viewdirs=True, use_camera=True,
        camera_model="pinhole_rot_noise_10k_rayo_rayd", cascade_level=1, cascade_samples="8", optim_autoexpo=False,
        basedir=out_dir, expname="exp", use_custom_optim=True, lrate=5e-4, non_linear_weight_decay=0.0,
        ckpt_path=None, no_reload=True, load_camera=False, load_test=False, add_ie=0, add_radial=0, add_od=0,
        grid_size=10, ray_o_noise_scale=1e-3, ray_d_noise_scale=1e-3, extrinsics_noise_scale=1.0,
        intrinsics_noise_scale=1.0, multiplicative_noise=True)
    info = {"intrinsics": spec["K_init"], "extrinsics": list(spec["poses"].numpy()), "H": Hh, "W": Ww}
    with emulated_device():
        start, models, cm = create_nerf("cpu", args, info)
        assert models["optim"].grad_sync is not None and models["grad_sync"] is models["optim"].grad_sync
        before = torch.cat([p.detach().reshape(-1).clone() for p in models["net_0"].parameters()])
        n = 4
        o, d, near = synth.nerfpp_rays(n, seed=5 + rank)            # every rank its own rays
        target = torch.rand(n, 3, generator=torch.Generator().manual_seed(9 + rank))
        optim = models["optim"]
        optim.zero_grad()
        far = TR.intersect_sphere(o, d)
        frac = torch.linspace(0, 1, 8)
        fg = near[:, None] + frac * (far - near)[:, None]
        bg = torch.linspace(0, 1, 8).expand(n, 8)
        ret = models["net_0"](o, d, far, fg, bg)
        loss = ((ret["rgb"] - target) ** 2).mean()
        loss.backward()
        local_grad = optim.flat_gradient().clone()
        optim.step()
        after = torch.cat([p.detach().reshape(-1) for p in models["net_0"].parameters()])
    np.save(os.path.join(out_dir, "after%d.npy" % rank), after.numpy())
    np.save(os.path.join(out_dir, "local_grad%d.npy" % rank), local_grad.numpy())
    np.save(os.path.join(out_dir, "moved%d.npy" % rank), np.array(float((after - before).abs().max())))
    dist.destroy_process_group()
