"""NeRF++ training-step rate (BASELINE config 5 shape: two cascade levels of 64 then +128 samples, foreground
and background networks): fwd + bwd of both levels on synthetic rays inside the unit sphere.  Prints one
JSON line (not the driver's bench)."""
import argparse
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(rays=2048, seed_offset=0, device="cuda"):
    """-> (step, modules, flop per step): one NeRF++ training step (nerfplusplus/ddp_train_nerf.py:491-550: two cascade levels,
    foreground + background networks, fwd + bwd) on this rank's synthetic rays; `modules` are what a gradient all-reduce
    covers (both NerfNet levels)."""
    from scnerf_amd import synthetic as synth
    from scnerf_amd.nerfplusplus import ddp_train_nerf as TR
    from scnerf_amd.nerfplusplus.ddp_model import NerfNet
    args = types.SimpleNamespace(max_freq_log2=10, max_freq_log2_viewdirs=4, netdepth=8, netwidth=256, use_viewdirs=True)
    torch.manual_seed(777)                                  # the same initial networks on every rank
    nets = [NerfNet(args).to(device), NerfNet(args).to(device)]
    n, s0, s1 = rays, 64, 128
    o, d, near = (t.to(device) for t in synth.nerfpp_rays(n, seed=5 + seed_offset))
    o.requires_grad_(True), d.requires_grad_(True)
    target = torch.rand(n, 3, generator=torch.Generator().manual_seed(9 + seed_offset)).to(device)
    steps_i = torch.arange(s0, dtype=torch.float32, device=device)

    def step(zero=True):
        if zero:
            for net in nets:
                for p in net.parameters():
                    p.grad = None
        o.grad = d.grad = None
        far = TR.intersect_sphere(o, d, check=False)
        st = (far - near) / (s0 - 1)
        # near + i * st for i = 0 .. s0 - 1 as ONE broadcast expression: the reference's script builds it as a python list of
        # 64 tensors (ddp_train_nerf.py:445-447: ~330 tiny launches per step with their backward, 2 ms of a 14 ms step here);
        # the values are the same bit for bit (i is exact in fp32, one multiplication and one addition either way)
        fg = TR.perturb_samples(near[:, None] + steps_i[None, :] * st[:, None])
        bg = TR.perturb_samples(torch.linspace(0., 1., s0, device=device).expand(n, s0))
        ret = nets[0](o, d, far, fg, bg)
        loss = ((ret["rgb"] - target) ** 2).mean()
        fg_s = TR.sample_pdf(.5 * (fg[..., 1:] + fg[..., :-1]), ret["fg_weights"].detach()[..., 1:-1], s1)
        fg1, _ = torch.sort(torch.cat((fg, fg_s), dim=-1))
        bg_s = TR.sample_pdf(.5 * (bg[..., 1:] + bg[..., :-1]), ret["bg_weights"].detach()[..., 1:-1], s1)
        bg1, _ = torch.sort(torch.cat((bg, bg_s), dim=-1))
        ret1 = nets[1](o, d, far, fg1, bg1)
        loss = loss + ((ret1["rgb"] - target) ** 2).mean()
        loss.backward()
        return loss
    mac_fg, mac_bg = 593408, 593408 + 2 * 256 * 21
    flop = n * (s0 + s0 + s1) * (mac_fg + mac_bg) * 2 * 3            # fwd + dgrad + wgrad
    return step, nets, flop


def run(rays=2048, steps=10, warmup=2):
    step, _, flop = build(rays)
    n = rays
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"metric": "rays/sec NeRF++ train-step (2 levels: 64 / 192 samples, fg + bg nets)",
            "value": n / dt, "unit": "rays/s", "rays": n, "steps": steps, "ms_per_step": dt * 1e3,
            "tflops_algorithmic": flop / dt / 1e12}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    print(json.dumps(run(a.rays, a.steps, a.warmup)))


if __name__ == "__main__":
    main()
