"""PSNR trajectory of a short training run on a procedural scene -- the second half of BASELINE.json's metric ("PSNR vs
reference").  The same loop on either side, from identical initial weights, ray batches and random numbers:

    python tools/psnr_trajectory.py --side oracle --steps 300 --out tests/golden/psnr_oracle.json      (CPU, any machine)
    python tools/psnr_trajectory.py --side gpu --arithmetic resident --steps 300 --out gpurun_out/psnr_resident.json

side "oracle": the torch-CPU restatement of the reference path (oracle/scnerf_oracle.py: render_rays + torch autograd +
the reference's Adam rule) -- the checker; side "gpu": scnerf_amd's render_rays + FusedAdam on the HIP kernels.
Loss as run_nerf.py:495-506 (img2mse of the fine and the coarse render), PSNR = mse2psnr (run_nerf_helpers.py:10-11)
of the fine render on held-out rays, every `--every` steps."""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from scnerf_amd import synthetic as synth      # noqa: E402

S_C, S_F, N_RAND, LR = 64, 64, 1024, 5e-4


def scene():
    rays = synth.procedural_rays()
    target = synth.procedural_targets(rays)
    g = torch.Generator().manual_seed(99)
    perm = torch.randperm(rays.shape[0], generator=g)
    held = perm[:2048]
    train = perm[2048:]
    return rays, target, train, held


def batch_of(step, train, n):
    g = torch.Generator().manual_seed(1000 + step)
    idx = train[torch.randint(0, train.shape[0], (n,), generator=g)]
    rnd = {"t_rand": torch.rand(n, S_C, generator=g), "u": torch.rand(n, S_F, generator=g)}
    return idx, rnd


def psnr_of(mse):
    return -10.0 * math.log(max(mse, 1e-20)) / math.log(10.0)


def run_oracle(steps, every, threads, perturb=0.0, perturb_seed=0):
    from oracle import scnerf_oracle as O        # checker side
    if threads:
        torch.set_num_threads(threads)
    rays, target, train, held = scene()
    pc = {k: v.clone().requires_grad_(True) for k, v in synth.network_params(seed=0).items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in synth.network_params(seed=1).items()}
    plist = list(pc.values()) + list(pf.values())
    if perturb:
        # every initial weight moved by a relative `perturb` (one fp32 rounding is 6e-8): how far apart do two runs of the
        # SAME fp32 algorithm end up?  That spread is the yardstick for any other fp32 implementation's trajectory.
        g = torch.Generator().manual_seed(perturb_seed)
        with torch.no_grad():
            for p in plist:
                p.mul_(1.0 + perturb * torch.randn(p.shape, generator=g))
    m = [torch.zeros_like(p) for p in plist]
    v = [torch.zeros_like(p) for p in plist]

    def evaluate():
        with torch.no_grad():
            out = O.render_rays(rays[held], pc, pf, S_C, S_F, None, None, None, None)
            return float(torch.mean((out["rgb_map"] - target[held]) ** 2))

    curve = [{"step": 0, "psnr": psnr_of(evaluate())}]
    t0 = time.time()
    for k in range(steps):
        idx, rnd = batch_of(k, train, N_RAND)
        for p in plist:
            p.grad = None
        out = O.render_rays(rays[idx], pc, pf, S_C, S_F, rnd["t_rand"], rnd["u"], None, None)
        loss = torch.mean((out["rgb_map"] - target[idx]) ** 2) + torch.mean((out["rgb0"] - target[idx]) ** 2)
        loss.backward()
        with torch.no_grad():
            O.adam_step([p for p in plist], [p.grad for p in plist], m, v, [k + 1] * len(plist), LR)
        if (k + 1) % every == 0 or k + 1 == steps:
            curve.append({"step": k + 1, "psnr": psnr_of(evaluate()), "loss": float(loss.detach())})
            print(curve[-1], "%.0f s" % (time.time() - t0), flush=True)
    return curve


def run_gpu(steps, every, arithmetic, perturb=0.0, perturb_seed=0, wgrad=None, return_nets=False):
    from scnerf_amd import ops
    from scnerf_amd.create_nerf import FusedNetworkQuery
    from scnerf_amd.optim import FusedAdam
    from scnerf_amd.render import render_rays
    from scnerf_amd.run_nerf_helpers import NeRF, get_embedder
    ops.mlp_arithmetic(arithmetic)
    ops.wgrad_arithmetic(wgrad or ("fp32" if arithmetic == "fp32" else "half"))
    dev = torch.device("cuda")
    rays, target, train, held = scene()
    rays_d, target_d = rays.to(dev), target.to(dev)

    gp = torch.Generator().manual_seed(perturb_seed)

    def make(seed):
        net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        sd = synth.network_params(seed=seed)
        if perturb:                                   # the same perturbation, in the same order, as run_oracle's
            sd = {k: v * (1.0 + perturb * torch.randn(v.shape, generator=gp)) for k, v in sd.items()}
        net.load_state_dict(sd)
        net = net.to(dev)
        net.flat_parameters()
        return net
    net_c, net_f = make(0), make(1)
    query = FusedNetworkQuery(get_embedder(10, 0)[0], get_embedder(4, 0)[0])
    opt = FusedAdam(list(net_c.parameters()) + list(net_f.parameters()), lr=LR)
    held_d = held.to(dev)

    def evaluate():
        with torch.no_grad():
            out = render_rays(rays_d[held_d], net_c, query, S_C, N_importance=S_F, network_fine=net_f, perturb=0.0,
                              raw_noise_std=0.0)
            return float(torch.mean((out["rgb_map"] - target_d[held_d]) ** 2))

    curve = [{"step": 0, "psnr": psnr_of(evaluate())}]
    for k in range(steps):
        idx, rnd = batch_of(k, train, N_RAND)
        idx = idx.to(dev)
        opt.zero_grad()
        out = render_rays(rays_d[idx], net_c, query, S_C, N_importance=S_F, network_fine=net_f, perturb=1.0,
                          raw_noise_std=0.0, _randoms={kk: vv.to(dev) for kk, vv in rnd.items()})
        loss = torch.mean((out["rgb_map"] - target_d[idx]) ** 2) + torch.mean((out["rgb0"] - target_d[idx]) ** 2)
        loss.backward()
        opt.step()
        if (k + 1) % every == 0 or k + 1 == steps:
            curve.append({"step": k + 1, "psnr": psnr_of(evaluate()), "loss": float(loss.detach())})
    if return_nets:
        return curve, (net_c, net_f)
    return curve


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", choices=("oracle", "gpu"), required=True)
    ap.add_argument("--arithmetic", default="resident")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--every", type=int, default=25)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--perturb", type=float, default=0.0, help="oracle side: relative perturbation of the initial weights")
    ap.add_argument("--perturb-seed", type=int, default=0)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    t0 = time.time()
    curve = (run_oracle(a.steps, a.every, a.threads, a.perturb, a.perturb_seed) if a.side == "oracle"
             else run_gpu(a.steps, a.every, a.arithmetic, a.perturb, a.perturb_seed))
    rec = {"side": a.side, "arithmetic": a.arithmetic if a.side == "gpu" else "torch-CPU fp32 (oracle)", "steps": a.steps,
           "n_rand": N_RAND, "samples": [S_C, S_F], "lr": LR, "scene": "procedural (scnerf_amd/synthetic.py), 24 views x 32 x 32, "
           "2048 held-out rays", "curve": curve, "final_psnr": curve[-1]["psnr"], "seconds": time.time() - t0,
           "perturb": a.perturb, "perturb_seed": a.perturb_seed}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({"final_psnr": rec["final_psnr"], "seconds": rec["seconds"]}))


if __name__ == "__main__":
    main()
