"""Headline benchmark: rays/s of one train step (forward + backward of render_rays, both networks,
64 coarse + 128 fine samples per ray) on N GPUs of one node -- BASELINE.json's metric/config.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rays 4096] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; every rank renders its own 4096 synthetic rays (weak scaling) and the flat
fp32 gradients of both networks are summed with ONE RCCL all-reduce per step.  Rank 0 prints one
JSON line.  `roofline` is measured live with HIP events around the dominant kernel's launches on
the stream they run on; `cpu_baseline` times the CPU oracle (a torch-CPU restatement of the
reference path, kind "port") on a bounded sample of the same workload on the box's host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE_FWD = 2 * 593408
S_C, S_F = 64, 128
PEAK_F32_MFMA_TFLOPS = 157.3              # MI355X_MICROARCH.md "Peak FP32 (matrix)"


def cpu_baseline(n_rays, iters=2):
    """The CPU oracle (torch-CPU restatement of the reference render_rays, kind "port") on the host
    cores of this box, forward + backward, on a bounded sample of the headline workload.  The
    intra-op thread count is chosen by a short probe (on many-core hosts torch's CPU kernels are
    fastest well below os.cpu_count()); the count actually used is reported as `cores`."""
    from oracle import scnerf_oracle as O            # checker only: the CPU leg of the report
    from scnerf_amd import synthetic as synth
    pc = {k: v.clone().requires_grad_(True) for k, v in synth.network_params(seed=0).items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in synth.network_params(seed=1).items()}

    def make_step(n):
        rays = synth.ray_batch(n, seed=1)
        target = synth.target_rgb(n, seed=2)
        rnd = synth.render_randoms(n, S_C, S_F, seed=3)

        def step():
            for d in (pc, pf):
                for v in d.values():
                    v.grad = None
            out = O.render_rays(rays, pc, pf, S_C, S_F, rnd["t_rand"], rnd["u"], rnd["noise_c"], rnd["noise_f"])
            loss = torch.mean((out["rgb_map"] - target) ** 2) + torch.mean((out["rgb0"] - target) ** 2)
            loss.backward()
        return step

    ncpu = os.cpu_count() or 1
    probe = make_step(128)
    best_t, best_thr = None, 1
    for thr in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128, ncpu)}):
        torch.set_num_threads(thr)
        probe()
        t0 = time.perf_counter()
        probe()
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, best_thr = dt, thr
        if dt > 4 * best_t:
            break
    torch.set_num_threads(best_thr)
    step = make_step(n_rays)
    step()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    best = min(ts)
    return {"value": n_rays / best, "unit": "rays/s", "cores": best_thr, "kind": "port",
            "sample": "%d rays x (64+128) samples, fwd+bwd, best of %d after 1 warm-up; oracle/scnerf_oracle.py "
                      "(torch-CPU fp32, anomaly detection off), %d intra-op threads chosen by probe on a %d-thread host"
                      % (n_rays, iters, best_thr, ncpu),
            "ms_per_step_sample": best * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=512)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from scnerf_amd import ops, synthetic as synth
    from scnerf_amd.create_nerf import FusedNetworkQuery
    from scnerf_amd.parallel import FlatGradAllReduce
    from scnerf_amd.render import render_rays
    from scnerf_amd.run_nerf_helpers import NeRF, get_embedder
    ops.check_layout()

    def make(seed):
        net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        net.load_state_dict(synth.network_params(seed=seed))
        return net.to(dev)

    net_c, net_f = make(0), make(1)
    query = FusedNetworkQuery(get_embedder(10, 0)[0], get_embedder(4, 0)[0])
    n = a.rays
    rays = synth.ray_batch(n, seed=1 + rank).to(dev)
    target = synth.target_rgb(n, seed=2 + rank).to(dev)
    net_c.flat_parameters(), net_f.flat_parameters()
    reducer = FlatGradAllReduce([net_c, net_f], world)
    inv = 1.0 / (3 * n)

    def step():
        reducer.zero()
        ret = render_rays(rays, net_c, query, S_C, retraw=True, perturb=1.0, N_importance=S_F,
                          network_fine=net_f, raw_noise_std=1.0)
        # loss = mse(rgb_map, target) + mse(rgb0, target): its gradient is fed to backward directly
        g1 = (ret["rgb_map"].detach() - target) * (2 * inv)
        g0 = (ret["rgb0"].detach() - target) * (2 * inv)
        torch.autograd.backward([ret["rgb_map"], ret["rgb0"]], [g1, g0])
        reducer.all_reduce()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    ops.PROFILE.reset(enabled=True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    ops.PROFILE.enabled = False
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / a.steps * 1e3

    if rank == 0:
        kern = ops.PROFILE.summary()
        # dominant kernel family by total time
        single = {k: v for k, v in kern.items() if not v["group"]}
        dom = max(single, key=lambda k: single[k]["total_ms"]) if single else None
        roof = None
        if dom:
            k = kern[dom]
            ach = k["flop_per_launch"] / (k["avg_ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                    "avg_launch_ms": k["avg_ms"], "launches_per_step": k["launches"] / a.steps,
                    "flop_per_launch": k["flop_per_launch"]}
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.isfile(pmc):
                roof["traffic"] = json.load(open(pmc)).get(dom)
        out = {
            "metric": "rays/sec (64+128 samples/ray) train-step", "value": n * world / (ms * 1e-3),
            "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: %d rays x (64 coarse + 128 fine), coarse+fine NeRF (D=8, W=256), "
                                   "fwd+bwd of render_rays per GPU, perturb=1, raw_noise_std=1" % n,
                       "rays_per_gpu": n, "parallelism": "ray-parallel x%d, 1 RCCL all-reduce/step" % world},
            "roofline": roof,
            "kernels": {k: {"avg_ms": v["avg_ms"], "launches_per_step": v["launches"] / a.steps,
                            "tflops": v["flop_per_launch"] / (v["avg_ms"] * 1e-3) / 1e12 if v["flop_per_launch"] else None}
                        for k, v in kern.items()},
            "step_flop_algorithmic": 3 * FLOP_PER_SAMPLE_FWD * (S_C + S_C + S_F) * n,
            "step_tflops": 3 * FLOP_PER_SAMPLE_FWD * (S_C + S_C + S_F) * n / (ms * 1e-3) / 1e12,
        }
        if world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(a.cpu_rays)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
