"""Headline benchmark: rays/s of one train step (forward + backward of render_rays, both networks,
64 coarse + 128 fine samples per ray) on N GPUs of one node -- BASELINE.json's metric/config.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rays 4096] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; every rank renders its own 4096 synthetic rays (weak scaling).  At N = 1 the step is
BASELINE.json's configs[1] exactly (precomputed rays of a fixed camera).  At N > 1 (or with --camera) the rays
of each step come from the learnable camera model (get_rays_kps_use_camera + NDC through the model: the ray
source of configs[2..3]) and ONE RCCL all-reduce per step sums the flat fp32 gradient buffer that holds both
networks AND the camera parameters -- the north star's collective; the extra work per step is two small
kernels, so per-N values stay comparable (N = 1 with the camera is reported under extras).  Rank 0 prints one
JSON line; at N = 1 `extras` adds short timings of the other configurations (camera curriculum states, PRD
loss, full-image inference, NeRF++).  `roofline` is measured live with HIP events around the dominant kernel's launches on
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
PEAK_BF16_MFMA_TFLOPS = 2500.0            # ibid., dense bf16
# the weight-gradient GEMMs in "split" arithmetic spend 6 bf16 MFMA products per fp32 product
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
# ... and 3 fp16 MFMA products (same matrix-pipe rate) where the "half" arithmetic applies
PEAK_HALF_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0


def _layer_region_peak(name):
    """ceiling of a `layer_split_kernel<8 layers[: n on three fp16 products]>` region: the eight layers' time at the
    matrix pipe's dense rate, n of them at three products per product and 8 - n at six"""
    import re
    m = re.search(r": (\d) on three fp16 products", name)
    n16 = int(m.group(1)) if m else 0
    return 8.0 / ((8 - n16) / PEAK_SPLIT_TFLOPS + n16 / PEAK_HALF_TFLOPS)


def cpu_baseline(n_rays, iters=3):
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
    ratio = ""
    pinned = os.path.join(ROOT, "profiles", "cpu_baseline_r02.json")
    if os.path.isfile(pinned):             # port vs the unmodified reference, timed side by side in the build container
        rec = json.load(open(pinned))
        ratio = "; port/reference time ratio pinned in profiles/cpu_baseline_r02.json (%d threads): %s" % (
            rec["threads"], ", ".join("%.2f at %s rays" % (v["port_over_reference_time"], k)
                                      for k, v in sorted(rec["sizes"].items(), key=lambda kv: int(kv[0]))))
    return {"value": n_rays / best, "unit": "rays/s", "cores": best_thr, "kind": "port",
            "sample": "%d rays x (64+128) samples, fwd+bwd, best of %d after 1 warm-up; oracle/scnerf_oracle.py "
                      "(torch-CPU fp32, anomaly detection off), %d intra-op threads chosen by probe on a %d-thread host%s"
                      % (n_rays, iters, best_thr, ncpu, ratio),
            "ms_per_step_sample": best * 1e3}


IMG_H, IMG_W, N_CAMS = 378, 504, 17      # LLFF 'fern' at factor 8: the image size / view count of configs[1..3]


def _kernel_table(kern, steps):
    """tflops = algorithmic fp32 FLOP / time; `peak` names the matrix-pipe ceiling of the arithmetic the entry
    runs in (the wgrad group: 8 of its 12 GEMMs, 87 % of its FLOPs, are the 256 x 256 ones)"""
    from scnerf_amd import ops
    split = ops.wgrad_arithmetic() == "split"
    out = {}
    for k, v in kern.items():
        e = {"avg_ms": v["avg_ms"], "launches_per_step": v["launches"] / steps,
             "tflops": v["flop_per_launch"] / (v["avg_ms"] * 1e-3) / 1e12 if v["flop_per_launch"] else None}
        if e["tflops"]:
            on_split = (split and k.startswith("wgrad(")) or "layer GEMMs" in k or k.startswith("layer_split_kernel")
            e["peak"] = PEAK_SPLIT_TFLOPS if on_split else PEAK_F32_MFMA_TFLOPS
            e["pipe"] = ("bf16 MFMA x6 (fp32 operands cut into 3 bf16, fp32 accumulate)" +
                         ("; encoding + layer 0 and the heads on the fp32 MFMA" if "layer GEMMs" in k else "")) if on_split else "fp32 MFMA"
            if k.startswith("layer_split_kernel") and "fp16" in k:
                e["peak"] = _layer_region_peak(k)
                e["pipe"] = ("fp16 MFMA x3 (fp32 operands scaled by a power of two per sample / layer and cut into 2 fp16, fp32 "
                             "accumulate) on the layers named, bf16 MFMA x6 on the others")
            elif "layer GEMMs" in k and ops.mlp_arithmetic() == "half":
                e["pipe"] = e["pipe"].replace("bf16 MFMA x6 (fp32 operands cut into 3 bf16, fp32 accumulate)",
                                              "fp16 MFMA x3 on 6 / 7 of the 8 layer GEMMs, bf16 MFMA x6 on the others")
        out[k] = e
    return out


def _timed(step, steps, warmup, sync, profile=True):
    """-> (ms per step, per-kernel table) of `steps` calls after `warmup` untimed ones"""
    from scnerf_amd import ops
    for _ in range(warmup):
        step()
    sync()
    ops.PROFILE.reset(enabled=profile)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    ops.PROFILE.enabled = False
    return dt / steps * 1e3, (_kernel_table(ops.PROFILE.summary(), steps) if profile else None)


def build_world(dev, rank, n):
    """networks, query object, camera model, per-rank synthetic data"""
    from scnerf_amd import synthetic as synth
    from scnerf_amd.create_nerf import FusedNetworkQuery
    from scnerf_amd.run_nerf_helpers import NeRF, get_embedder

    def make(seed):
        net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        net.load_state_dict(synth.network_params(seed=seed))
        net = net.to(dev)
        net.flat_parameters()
        return net
    cam, _ = synth.camera_model(IMG_H, IMG_W, n_cams=N_CAMS, seed=4)
    g = torch.Generator().manual_seed(100 + rank)
    kps = torch.stack([torch.randint(0, IMG_W, (n,), generator=g), torch.randint(0, IMG_H, (n,), generator=g)], -1)
    idx = torch.randint(0, N_CAMS, (n,), generator=g)
    return dict(net_c=make(0), net_f=make(1), cam=cam.to(dev),
                query=FusedNetworkQuery(get_embedder(10, 0)[0], get_embedder(4, 0)[0]),
                rays=synth.ray_batch(n, seed=1 + rank).to(dev), target=synth.target_rgb(n, seed=2 + rank).to(dev),
                kps=kps.to(dev), idx=idx.to(dev), n=n)


def fixed_camera_step(w, reducer):
    """configs[1]: precomputed rays of a fixed camera -> render_rays fwd + bwd (+ the all-reduce)"""
    from scnerf_amd.render import render_rays
    inv = 1.0 / (3 * w["n"])

    def step():
        reducer.zero()
        ret = render_rays(w["rays"], w["net_c"], w["query"], S_C, retraw=True, perturb=1.0, N_importance=S_F,
                          network_fine=w["net_f"], raw_noise_std=1.0)
        # loss = mse(rgb_map, target) + mse(rgb0, target): its gradient is fed to backward directly
        g1 = (ret["rgb_map"].detach() - w["target"]) * (2 * inv)
        g0 = (ret["rgb0"].detach() - w["target"]) * (2 * inv)
        torch.autograd.backward([ret["rgb_map"], ret["rgb0"]], [g1, g0])
        reducer.all_reduce()
    return step


def learnable_camera_step(w, reducer):
    """configs[2..3] ray source: key-point rays through the learnable camera model, NDC through its intrinsics,
    render fwd + bwd down to the camera parameters (+ the all-reduce over networks AND camera)"""
    from scnerf_amd.get_rays import get_rays_kps_use_camera
    from scnerf_amd.render import render
    inv = 1.0 / (3 * w["n"])
    kw = dict(network_fn=w["net_c"], network_fine=w["net_f"], network_query_fn=w["query"], N_samples=S_C,
              N_importance=S_F, perturb=1.0, raw_noise_std=1.0, use_viewdirs=True, white_bkgd=False, near=0., far=1.)

    def step():
        reducer.zero()
        rays_o, rays_d = get_rays_kps_use_camera(H=IMG_H, W=IMG_W, camera_model=w["cam"],
                                                 idx_in_camera_param=w["idx"], kps_list=w["kps"])
        rgb, _, _, extras = render(H=IMG_H, W=IMG_W, chunk=1 << 15, rays=torch.stack([rays_o, rays_d]), retraw=True,
                                   camera_model=w["cam"], mode="train", **kw)
        g1 = (rgb.detach() - w["target"]) * (2 * inv)
        g0 = (extras["rgb0"].detach() - w["target"]) * (2 * inv)
        torch.autograd.backward([rgb, extras["rgb0"]], [g1, g0])
        reducer.all_reduce()
    return step


def extras_single_gpu(w, dev, sync):
    """Short timings of the configurations that are not the headline: each a few steps."""
    from scnerf_amd import synthetic as synth
    from scnerf_amd.get_rays import get_rays_kps_use_camera
    from scnerf_amd.parallel import FlatGradAllReduce
    from scnerf_amd.ray_dist_loss import proj_ray_dist_loss_single
    import types
    out = {}
    cam = w["cam"]
    groups = {"ie": (cam.intrinsics_noise, cam.extrinsics_noise), "od": (cam.ray_o_noise, cam.ray_d_noise)}
    states = {}
    for name, on in (("none", ()), ("ie", ("ie",)), ("ie+od", ("ie", "od"))):
        for gname, tensors in groups.items():
            for t_ in tensors:
                t_.requires_grad_(gname in on)
                t_.grad = None
        red = FlatGradAllReduce([w["net_c"], w["net_f"], cam], 1)
        ms, kern = _timed(learnable_camera_step(w, red), 5, 2, sync)
        states[name] = {"ms_per_step": ms, "rays_per_s": w["n"] / (ms * 1e-3), "flat_gradient_floats": int(red.flat.numel()),
                        "kernels": kern}
    out["config2_camera_curriculum"] = {"workload": "configs[2]: %d rays x (64+128), rays from the learnable camera model "
                                        "(%d views, %dx%d), fwd+bwd incl. camera gradients" % (w["n"], N_CAMS, IMG_H, IMG_W),
                                        "states": states}
    # configs[3]: the projected-ray-distance term of one image pair, forward + backward into the camera parameters
    E = cam.get_extrinsic().detach().cpu()
    K = cam.get_intrinsic().detach().cpu()
    k0, k1 = synth.matched_keypoints(IMG_H, IMG_W, K, E[0], E[1], 1024, seed=8)
    k0, k1 = k0.to(dev), k1.to(dev)
    args = types.SimpleNamespace(proj_ray_dist_threshold=5.0)
    i_map = torch.arange(N_CAMS).numpy()

    def prd_step():
        for p in cam.parameters():
            p.grad = None
        r0 = get_rays_kps_use_camera(H=IMG_H, W=IMG_W, camera_model=cam, idx_in_camera_param=0, kps_list=k0)
        r1 = get_rays_kps_use_camera(H=IMG_H, W=IMG_W, camera_model=cam, idx_in_camera_param=1, kps_list=k1)
        loss, _ = proj_ray_dist_loss_single(kps0_list=k0, kps1_list=k1, img_idx0=0, img_idx1=1, rays0=r0, rays1=r1,
                                            mode="train", device=dev, H=IMG_H, W=IMG_W, args=args, camera_model=cam,
                                            method="NeRF", i_map=i_map)
        loss.backward()
    ms, _ = _timed(prd_step, 20, 3, sync, profile=False)
    out["config3_prd_term"] = {"workload": "projected-ray-distance loss of one image pair, 1024 matches: 2 x camera rays + "
                               "loss fwd + bwd into the camera parameters (incl. the .item() the API returns)",
                               "ms_per_call": ms}
    from tools import bench_infer, bench_nerfpp
    out["full_image_inference"] = bench_infer.run(images=3)
    out["config5_nerfpp"] = bench_nerfpp.run(rays=2048, steps=5, warmup=2)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=512)
    ap.add_argument("--camera", action="store_true", help="rays from the learnable camera model also at N = 1")
    ap.add_argument("--mlp-arithmetic", choices=("split", "fp32", "half"), default=None,
                    help="the eight 256-wide layers of the training forward and of the data-gradient chain: GEMMs over all "
                         "samples on the 16-bit matrix pipe -- 'half' (default): three fp16 products where the input carries "
                         "per-sample maxima, six bf16 products elsewhere; 'split': six bf16 products everywhere -- or "
                         "'fp32': inside the fused fp32-MFMA kernels")
    ap.add_argument("--wgrad-arithmetic", choices=("split", "fp32"), default=None,
                    help="256 x 256 weight-gradient GEMMs: bf16 matrix pipe with exactly cut fp32 operands (default) "
                         "or the exact-fp32 MFMA")
    ap.add_argument("--backend", default=os.environ.get("SCNERF_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend (nccl = RCCL; gloo for a functional check of N ranks on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="all ranks on cuda:0 (functional check, with --backend gloo)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if a.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (a.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    from scnerf_amd import ops
    from scnerf_amd.parallel import FlatGradAllReduce
    ops.check_layout()
    if a.wgrad_arithmetic:
        ops.wgrad_arithmetic(a.wgrad_arithmetic)
    if a.mlp_arithmetic:
        ops.mlp_arithmetic(a.mlp_arithmetic)
    n = a.rays
    w = build_world(dev, rank, n)
    with_camera = a.camera or world > 1
    if with_camera:
        reducer = FlatGradAllReduce([w["net_c"], w["net_f"], w["cam"]], world)
        step = learnable_camera_step(w, reducer)
    else:
        reducer = FlatGradAllReduce([w["net_c"], w["net_f"]], world)
        step = fixed_camera_step(w, reducer)

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
            split_kernel = dom.startswith("layer_split_kernel")
            peak = (_layer_region_peak(dom) if split_kernel else PEAK_F32_MFMA_TFLOPS)
            roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": peak,
                    "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                    "avg_launch_ms": k["avg_ms"], "launches_per_step": k["launches"] / a.steps,
                    "flop_per_launch": k["flop_per_launch"]}
            if split_kernel:
                roof["peak_note"] = ("fp32 products per second; peak = dense bf16 / fp16 MFMA rate (2500) over the partial "
                                     "products per fp32 product: 6 (three-way bf16 cut) on the layers that run on bf16, 3 "
                                     "(two-way fp16 cut) on those the kernel name counts -- 416.7 for a pass all on bf16")
                roof["achieved_over_fp32_mfma_peak"] = ach / PEAK_F32_MFMA_TFLOPS
                roof["measured"] = ("HIP events around each of the launches of this size inside the timed region: one launch "
                                    "runs the eight 256-wide layers of a pass (forward layers 1-8, or the eight transposed "
                                    "layers of the data-gradient chain); flop_per_launch is the mean of the two")
                roof["note"] = ("power-bound: matrix pipe busy 68-70 % of the cycles at a shader clock of 1.7 GHz, against the "
                                "2.4 GHz the peak is quoted at (clock read inside the kernel, profiles/r02f_layer_split_lab.txt; "
                                "counters, profiles/r02f_pmc_kernels.txt; DESIGN.md 4.2b): at the clock it is given the kernel "
                                "delivers frac x 2.4 / 1.73 of the matrix pipe's rate")
                if "fp16" in dom:
                    roof["note"] = ("six (forward) / seven (data gradients) of the eight layers on three fp16 products: half the "
                                    "matrix-pipe work of the all-bf16 pass -- the launch is bound by its non-MFMA work now "
                                    "(epilogue 26 %, loads / cuts / scalar work in the slab loop 27 % of the cycles of an fp16 "
                                    "layer; profiles/r02g_layer_split_lab.txt, DESIGN.md 4.2b)")
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.isfile(pmc):
                rec = json.load(open(pmc))
                roof["traffic"] = rec.get(dom)
                roof["traffic_source"] = rec.get("_source", "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command)")
        source = ("rays from the learnable camera model (%d views, %dx%d; configs[2..3] ray source), camera parameters "
                  "in the all-reduced flat buffer" % (N_CAMS, IMG_H, IMG_W)) if with_camera else \
            "precomputed rays of a fixed camera"
        out = {
            "metric": "rays/sec (64+128 samples/ray) train-step", "value": n * world / (ms * 1e-3),
            "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "arithmetic": {
                "training forward and data gradients, the eight 256-wide layers": (
                    "per-layer GEMMs, fp32 operands cut EXACTLY into 3 bf16 numbers each (weights once per step, "
                    "activations / gradients in registers), 6 of the 9 partial products on v_mfma_f32_32x32x16_bf16, fp32 "
                    "accumulate; per-layer error vs fp64 1.3x the fp32 MFMA's rms (profiles/parity_r02.json "
                    "layer_gemm_arithmetic_*; --mlp-arithmetic fp32 keeps them inside the fused fp32-MFMA kernels)")
                if ops.mlp_arithmetic() == "split" else (
                    "GEMMs over all samples; forward layers 2-4 and 6-8 and the data-gradient layers 7^T .. 1^T: fp32 operands "
                    "scaled by a power of two (per sample from the maxima the producing layer leaves, per layer for the "
                    "weights) and cut into 2 fp16 numbers, 3 partial products on v_mfma_f32_32x32x16_f16, fp32 accumulate; "
                    "layer 1, the skip layer and feature_linear^T: 3 bf16 numbers, 6 products (--mlp-arithmetic split: "
                    "all of them); per-layer error vs fp64 that of the fp32 MFMA (profiles/parity_r02.json "
                    "layer_gemm_arithmetic_*)")
                if ops.mlp_arithmetic() == "half" else "fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32 (fused kernels)",
                "encoding + layer 0, heads, inference forward, narrow weight gradients": "fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32",
                "256x256 weight gradients": (
                    "fp32 operands cut EXACTLY into 3 bf16 numbers each, 6 of the 9 partial products on "
                    "v_mfma_f32_32x32x16_bf16, fp32 accumulate; error vs fp64 = the fp32 MFMA kernel's "
                    "(profiles/parity_r02.json wgrad256_arithmetic_*; --wgrad-arithmetic fp32 selects the latter)")
                if ops.wgrad_arithmetic() == "split" else "fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32"},
            "data": "synthetic",
            "config": {"workload": "configs[1]: %d rays x (64 coarse + 128 fine), coarse+fine NeRF (D=8, W=256), "
                                   "fwd+bwd of render_rays per GPU, perturb=1, raw_noise_std=1; %s" % (n, source),
                       "rays_per_gpu": n,
                       "parallelism": "ray-parallel x%d, 1 %s all-reduce/step of %d floats" % (
                           world, "RCCL" if a.backend == "nccl" else a.backend, int(reducer.flat.numel()))},
            "roofline": roof,
            "kernels": _kernel_table(kern, a.steps),
            "step_flop_algorithmic": 3 * FLOP_PER_SAMPLE_FWD * (S_C + S_C + S_F) * n,
            "step_tflops": 3 * FLOP_PER_SAMPLE_FWD * (S_C + S_C + S_F) * n / (ms * 1e-3) / 1e12,
        }
        if world == 1 and not a.no_extras:
            out["extras"] = extras_single_gpu(w, dev, sync)
        if world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(a.cpu_rays)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
