"""Build container only: times the UNMODIFIED reference render_rays (NeRF/render.py via oracle/ref_import.py)
and the CPU oracle (oracle/scnerf_oracle.py, what bench.py's cpu_baseline leg times on the GPU box, where the
reference cannot travel) side by side -- same rays, same network weights, same thread count, forward + backward,
best of >= 3 timed iterations after one warm-up -- and writes
profiles/cpu_baseline_r02.json with the port / reference ratio per size.

    python tools/cpu_port_vs_reference.py [--threads 8] [--iters 3] [--sizes 512 4096]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S_C, S_F = 64, 128


def best_of(fn, iters):
    fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--sizes", type=int, nargs="+", default=[512, 4096])
    a = ap.parse_args()
    from oracle import scnerf_oracle as O
    from oracle.gen_golden import ref_network
    from oracle.ref_import import load_reference
    from scnerf_amd import synthetic as synth
    ns = load_reference()
    torch.autograd.set_detect_anomaly(False)       # the reference switches it on at import (run_nerf_helpers.py:7)
    torch.set_num_threads(a.threads)
    out = {"threads": a.threads, "iters": a.iters, "host": os.uname().nodename, "torch": torch.__version__,
           "workload": "render_rays fwd + bwd, %d coarse + %d fine samples, perturb = 1, raw_noise_std = 1" % (S_C, S_F),
           "sizes": {}}
    for n in a.sizes:
        rays = synth.ray_batch(n, seed=1)
        target = synth.target_rgb(n, seed=2)
        rnd = synth.render_randoms(n, S_C, S_F, seed=3)
        net_c, query = ref_network(ns, synth.network_params(seed=0), S_F)
        net_f, _ = ref_network(ns, synth.network_params(seed=1), S_F)

        def ref_step():
            for m in (net_c, net_f):
                m.zero_grad(set_to_none=True)
            ret = ns.render.render_rays(rays, net_c, query, S_C, retraw=True, perturb=1.0, N_importance=S_F,
                                        network_fine=net_f, raw_noise_std=1.0)
            (torch.mean((ret["rgb_map"] - target) ** 2) + torch.mean((ret["rgb0"] - target) ** 2)).backward()
        pc = {k: v.clone().requires_grad_(True) for k, v in synth.network_params(seed=0).items()}
        pf = {k: v.clone().requires_grad_(True) for k, v in synth.network_params(seed=1).items()}

        def port_step():
            for d in (pc, pf):
                for v in d.values():
                    v.grad = None
            o = O.render_rays(rays, pc, pf, S_C, S_F, rnd["t_rand"], rnd["u"], rnd["noise_c"], rnd["noise_f"])
            (torch.mean((o["rgb_map"] - target) ** 2) + torch.mean((o["rgb0"] - target) ** 2)).backward()
        t_ref, all_ref = best_of(ref_step, a.iters)
        t_port, all_port = best_of(port_step, a.iters)
        out["sizes"][str(n)] = {"reference_s": t_ref, "port_s": t_port, "reference_rays_per_s": n / t_ref,
                                "port_rays_per_s": n / t_port, "port_over_reference_time": t_port / t_ref,
                                "reference_all_s": all_ref, "port_all_s": all_port}
        print(n, out["sizes"][str(n)], flush=True)
    with open(os.path.join(ROOT, "profiles", "cpu_baseline_r02.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
