"""Headline benchmark: rays/s of one train step (forward + backward of render_rays, both networks,
64 coarse + 128 fine samples per ray) on N GPUs of one node -- BASELINE.json's metric/config.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rays 4096] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; every rank renders its own 4096 synthetic rays (weak scaling).  The workload is the same at every N
(`--config`, default 1): 1 = BASELINE.json's configs[1] (precomputed rays of a fixed camera; the all-reduce carries both
networks' gradients), 2 = configs[2] (rays from the learnable camera model, NDC through its intrinsics; the all-reduce
carries networks AND camera parameters), 3 = configs[3] (the same + the projected-ray-distance term of one image pair in
every step, NeRF/run_nerf.py:508-598), 4 = configs[4] (NeRF++: two cascade levels, foreground + background networks,
nerfplusplus/ddp_train_nerf.py:491-550).  At N > 1 ONE all-reduce per step (RCCL) sums the flat fp32 gradient buffer; at
N = 1 there is no collective.  Rank 0 prints ONE compact JSON
line (< 4 KB: the contract's keys, `roofline`, `cpu_baseline`, the all-fp32-MFMA step; tests/test_bench_line.py) and writes
the full record -- per-kernel tables, and at N = 1 short timings of the other configurations (camera curriculum states, PRD
loss, full-image inference, NeRF++) -- to profiles/bench_detail_n<N>.json, named in the line.  `--gpus N` without a launcher
starts its own N ranks.  `roofline` is measured live with HIP events around the dominant kernel's launches on the stream they
run on, its `traffic` comes from the newest PMC summary whose source hash matches the kernels (tools/round_evidence.sh);
`cpu_baseline` times the UNMODIFIED reference render_rays at 4096 rays x (64 + 128) on the host cores where its tree is
present (the build container; kind "reference"), else -- on the GPU box, where nothing of the reference travels -- the CPU
oracle that is pinned to it, on a bounded sample (kind "port"; tools/cpu_port_vs_reference.py relates the two here).
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
PEAK_16BIT_MFMA_TFLOPS = 2500.0           # ibid., dense bf16 / fp16
PEAK_HBM_GBS = 8000.0                     # ibid., HBM3E spec (6290 measured with a float4 copy)
MEASURED_HBM_GBS = 6290.0


def csrc_sha16():
    """identity of the kernel sources: a counter summary taken at other sources is refused as stale"""
    import hashlib
    d = os.path.join(ROOT, "scnerf_amd", "csrc")
    h = hashlib.sha256()
    for sub in ("", "device"):
        dd = os.path.join(d, sub)
        for f in sorted(os.listdir(dd)):
            if f.endswith((".hip", ".h")):
                h.update(f.encode() + b"\0" + open(os.path.join(dd, f), "rb").read())
    return h.hexdigest()[:16]


def region_model(name):
    """What a timed region must do at least, from its name: algorithmic fp32 FLOP, matrix-pipe products per fp32
    product (None: the exact-fp32 MFMA), algorithmic HBM bytes per launch.  DESIGN.md section 5 states the per-sample
    figures: forward 2 x 593 408 FLOP; training forward writes the activation workspace ((2528 + 72) floats per sample
    for 3-D points) + raw, reads the point; the data-gradient chain writes the gradient workspace (2432 floats), reads
    the ReLU bit words and d raw; the weight gradients read both workspaces."""
    import re
    from scnerf_amd import mlp_layout as ML
    m = re.search(r"/P=(\d+)", name)
    if not m:
        return None
    P = int(m.group(1))
    pd = 4 if "/pd4" in name else 3
    lay = ML.layout(pd)
    mac = 593408 + (2 * 256 * 21 if pd == 4 else 0)
    save_b = (lay.save_floats_per_sample + ML.MASK_WORDS_PER_SAMPLE) * 4
    grad_b = ML.GRAD_FLOATS_PER_SAMPLE * 4
    fwd_io = 4 * pd + 16                               # the point in, raw out
    if name.startswith("mlp_fwd_h3_kernel") or name.startswith("mlp_fwd_kernel"):
        products = 3 if "_h3_" in name else None
        # (the fused stages also write the depths and points of their samples: 4 + 12 bytes; the fine stage reads no points)
        b = fwd_io + (save_b if "/train" in name else 0) + (16 if ("coarse stage" in name or "fine stage" in name) else 0)
        return dict(flop=2 * mac * P, products=products, bytes=b * P)
    if name.startswith("mlp_bwd_h3_kernel") or name.startswith("mlp_bwd_kernel"):
        return dict(flop=2 * mac * P, products=3 if "_h3_" in name else None,
                    bytes=(grad_b + ML.MASK_WORDS_PER_SAMPLE * 4 + 16 + 4 * pd + 4 * pd + 12) * P)
    if name.startswith("wgrad256_kernel"):
        return dict(flop=8 * 2 * 256 * 256 * P, products=3 if "half" in name else None,
                    bytes=16 * 256 * 4 * P)
    if name.startswith("wgrad("):
        # 8 of the 12 GEMMs (87 % of the FLOP) are the 256 x 256 ones; on three fp16 products the shapes with a
        # tile-native dZ (2 x 256 x encoded point, 128 x (256 + 32)) join them (csrc/wgrad_half_narrow.h); the rest
        # (rgb rows, alpha_linear's vector-matrix product) is HBM-rate VALU work, priced at the fp32 rate
        big = 8 * 2 * 256 * 256 * P
        return dict(flop=2 * mac * P, products=None, bytes=(lay.save_floats_per_sample * 4 + grad_b) * P,
                    mixed=big, mixed_half=big + 2 * (2 * 256 * lay.e_width + 128 * 256 + 128 * 32) * P)
    return None


def floors(name, avg_ms, wgrad_products=3):
    """both floors of a region and which one binds: MFMA (FLOP x products / dense 16-bit rate, or FLOP / the fp32 MFMA
    rate) and HBM (algorithmic bytes / 8 TB/s)"""
    md = region_model(name)
    if md is None or not avg_ms:
        return None
    if md.get("mixed") and wgrad_products:
        f16 = min(md["flop"], md["mixed_half"])
        t_mfma = f16 * wgrad_products / (PEAK_16BIT_MFMA_TFLOPS * 1e12) + (md["flop"] - f16) / (PEAK_F32_MFMA_TFLOPS * 1e12)
        pipe = "fp16 MFMA x3 on the eight 256 x 256 GEMMs and the narrow ones with a tile-native dZ, fp32 on the rest"
    elif md["products"]:
        t_mfma = md["flop"] * md["products"] / (PEAK_16BIT_MFMA_TFLOPS * 1e12)
        pipe = "fp16 MFMA x%g (fp32 operands cut into two fp16 planes, fp32 accumulate)" % md["products"]
    else:
        t_mfma = md["flop"] / (PEAK_F32_MFMA_TFLOPS * 1e12)
        pipe = "fp32 MFMA"
    t_hbm = md["bytes"] / (PEAK_HBM_GBS * 1e9)
    t = avg_ms * 1e-3
    return {"pipe": pipe, "flop_per_launch": md["flop"], "algorithmic_bytes_per_launch": md["bytes"],
            "floor_mfma_ms": t_mfma * 1e3, "floor_hbm_ms": t_hbm * 1e3,
            "bound": "mfma" if t_mfma >= t_hbm else "hbm", "frac_mfma": t_mfma / t, "frac_hbm": t_hbm / t,
            "frac": max(t_mfma, t_hbm) / t,
            "tflops": md["flop"] / t / 1e12, "gbytes_per_s": md["bytes"] / t / 1e9}


def _time_steps(step, iters):
    step()                                     # one warm-up
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def cpu_baseline_reference(n_rays=4096, iters=3):
    """BASELINE.md section 2 by the book: the UNMODIFIED reference `render_rays` (NeRF/render.py:186-300 with its
    raw2outputs / sample_pdf, create_nerf.run_network, run_nerf_helpers.NeRF + Embedder) imported from the reference tree
    (oracle/ref_ship.py: $SCNERF_REFERENCE_ROOT or the build container's tree) on CPU tensors: `n_rays` x (64 + 128), forward + loss.backward(), 1 warm-up + `iters` timed,
    best and median, anomaly detection off -- plus one figure with it on, as the reference ships
    (run_nerf_helpers.py:7), at 1024 rays.  About 80 s of CPU work.  None when no reference tree is here."""
    from oracle import ref_ship
    if ref_ship.ensure() is None:
        return None
    from oracle.gen_golden import ref_network
    from oracle.ref_import import load_reference
    from scnerf_amd import synthetic as synth
    ns = load_reference()
    torch.autograd.set_detect_anomaly(False)       # the reference switches it on at import
    net_c, query = ref_network(ns, synth.network_params(seed=0), S_F)
    net_f, _ = ref_network(ns, synth.network_params(seed=1), S_F)

    def make_step(n):
        rays = synth.ray_batch(n, seed=1)
        target = synth.target_rgb(n, seed=2)

        def step():
            for m in (net_c, net_f):
                m.zero_grad(set_to_none=True)
            ret = ns.render.render_rays(rays, net_c, query, S_C, retraw=True, perturb=1.0, N_importance=S_F,
                                        network_fine=net_f, raw_noise_std=1.0)
            (torch.mean((ret["rgb_map"] - target) ** 2) + torch.mean((ret["rgb0"] - target) ** 2)).backward()
        return step

    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    # thread count: one 1024-ray step per candidate (large enough that the choice carries over to 4096 rays: a probe on
    # 256 rays picked 8 threads on a 256-thread EPYC where 16-32 are faster at the headline size); the candidates stop at
    # 64 -- torch's CPU kernels only lose beyond that on this path, and one oversubscribed step takes minutes
    small = make_step(1024)
    cands = sorted({min(ncpu, c) for c in (8, 16, 32, 64)})
    torch.set_num_threads(cands[0])
    small()                                         # warm-up (allocator, thread pool)
    probe = {}
    for thr in cands:
        torch.set_num_threads(thr)
        t0 = time.perf_counter()
        small()
        probe[thr] = time.perf_counter() - t0
        if probe[thr] > 2 * min(probe.values()):
            break
    thr = min(probe, key=probe.get)
    torch.set_num_threads(thr)
    torch.autograd.set_detect_anomaly(True)
    try:
        t0 = time.perf_counter()
        small()
        t_anomaly = time.perf_counter() - t0
    finally:
        torch.autograd.set_detect_anomaly(False)
    best, median = _time_steps(make_step(n_rays), iters)
    torch.set_num_threads(default_threads)
    return {"value": n_rays / best, "unit": "rays/s", "cores": thr, "kind": "reference",
            "sample": "unmodified reference render_rays (NeRF/render.py:186-300), %d rays x (64+128), fwd+bwd, torch-CPU fp32, "
                      "anomaly detection off, 1 warm-up + %d timed; %d intra-op threads (fastest of %s on a 1024-ray step) on a "
                      "host with os.cpu_count() = %d" % (n_rays, iters, thr, "/".join(str(c) for c in probe), ncpu),
            "best_ms": best * 1e3, "median_ms": median * 1e3, "value_median": n_rays / median,
            "os_cpu_count": ncpu, "threads": thr,
            "rays_per_s_1024_rays_by_threads": {str(k): 1024 / v for k, v in probe.items()},
            "rays_per_s_1024_rays_anomaly_on_as_shipped": 1024 / t_anomaly}


def cpu_baseline_port(n_rays, iters=3):
    """Fallback where no reference tree is present: the CPU oracle (torch-CPU restatement of the reference render_rays,
    kind "port") on the host cores of this box, forward + backward, on a bounded sample of the headline workload."""
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
    default_threads = torch.get_num_threads()
    # thread count as the reference leg picks it: one 1024-ray step per candidate (a probe on 128 rays picked 16 threads on a
    # 256-thread EPYC where 32 are faster at the headline size)
    small = make_step(1024)
    cands = sorted({min(ncpu, c) for c in (8, 16, 32, 64)})
    torch.set_num_threads(cands[0])
    small()
    probe = {}
    for thr in cands:
        torch.set_num_threads(thr)
        t0 = time.perf_counter()
        small()
        probe[thr] = time.perf_counter() - t0
        if probe[thr] > 2 * min(probe.values()):
            break
    thr = min(probe, key=probe.get)
    torch.set_num_threads(thr)
    best, median = _time_steps(make_step(n_rays), iters)
    torch.set_num_threads(default_threads)
    return {"value": n_rays / best, "unit": "rays/s", "cores": thr, "kind": "port",
            "sample": "no reference tree on this machine (nothing of the Python reference travels to the GPU box): "
                      "oracle/scnerf_oracle.py, the torch-CPU fp32 restatement pinned to the reference's goldens (same speed as "
                      "the unmodified reference within 3 %%: profiles/cpu_baseline_r02.json), %d rays x (64+128) -- a bounded "
                      "sample of the 4096-ray step --, fwd+bwd, anomaly detection off, 1 warm-up + %d timed; %d intra-op threads "
                      "(fastest of %s on a 1024-ray step) on a host with os.cpu_count() = %d"
                      % (n_rays, iters, thr, "/".join(str(c) for c in probe), ncpu),
            "best_ms": best * 1e3, "median_ms": median * 1e3, "value_median": n_rays / median,
            "os_cpu_count": ncpu, "threads": thr,
            "rays_per_s_1024_rays_by_threads": {str(k): 1024 / v for k, v in probe.items()}}


def cpu_baseline(n_rays=4096, port_rays=2048):
    """the unmodified reference where its tree is (build container), else the pinned port on a bounded sample: ~25 s of CPU
    work on the GPU box's host (4 probe steps of 1024 rays + 3 steps of 2048 rays at ~300 rays/s)"""
    return cpu_baseline_reference(n_rays) or cpu_baseline_port(port_rays, iters=2)


METRIC = "rays/sec (64+128 samples/ray) train-step"


def metric_for(cfg):
    """the metric label of `--config k` (the headline's label is BASELINE.json's; the other configurations say what they time)"""
    return {1: METRIC, 2: METRIC + " (configs[2]: rays from the learnable camera model)",
            3: METRIC + " (configs[3]: learnable camera + projected-ray-distance term)",
            4: "rays/sec NeRF++ train-step (configs[4]: two cascade levels x fg + bg network)"}[cfg]


class Guard:
    """First contact with RCCL at N > 1 must not hang the driver: every stage of the run that can block on another rank
    (rendezvous, communicator start-up, the first collective, the step loops) runs under a deadline.  A deadline that
    passes, an exception, or the launcher's SIGTERM (another rank died) makes rank 0 print ONE JSON line with an `error`
    field -- the contract's keys, `value` null -- and the process exits non-zero; other ranks say the same on stderr."""

    def __init__(self, json_out, rank, world, args):
        self.json_out, self.rank, self.world, self.args = json_out, rank, world, args
        self.stage, self.done = "start", False
        import threading
        self._lock = threading.Lock()

    def error_line(self, stage, err):
        a = self.args
        return json.dumps({"metric": metric_for(getattr(a, "config", 1)), "value": None, "unit": "rays/s", "n_gpus": self.world,
                           "steps": a.steps, "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True,
                           "scaling": "weak", "vs_baseline": None, "data": "synthetic", "error": str(err)[:1500],
                           "stage": stage, "rank": self.rank, "backend": a.backend})

    def fail(self, stage, err, code=3):
        with self._lock:
            if not self.done:
                self.done = True
                text = self.error_line(stage, err)
                sys.stderr.write("[bench rank %d] FAILED in stage '%s': %s\n" % (self.rank, stage, err))
                sys.stderr.flush()
                if self.rank == 0:
                    self.json_out.write(text + "\n")
                    self.json_out.flush()
        os._exit(code)            # (not sys.exit: a thread blocked inside a collective never returns to unwind)

    def deadline(self, stage, seconds):
        guard = self

        class _D:
            def __enter__(self_):
                import threading
                guard.stage = stage
                self_.t = threading.Timer(seconds, lambda: guard.fail(stage, "no progress within %.0f s (deadline of this stage; "
                                                                      "SCNERF_BENCH_TIMEOUT_SCALE stretches all of them)" % seconds))
                self_.t.daemon = True
                self_.t.start()

            def __exit__(self_, et, ev, tb):
                self_.t.cancel()
                return False
        return _D()

    def install_sigterm(self):
        """SIGTERM (torch.distributed.run sends it to the surviving ranks when one dies) is answered by a thread of its own:
        a Python-level handler only runs once the main thread returns from the C call it is blocked in -- which, inside a
        collective whose peer is gone, is never.  The interpreter's C-level handler writes the signal number to the wake-up
        descriptor at once, in whichever thread the kernel picked; the watcher thread reads it there."""
        import signal
        import socket
        import threading
        try:
            r, w = socket.socketpair()
            w.setblocking(False)
            signal.signal(signal.SIGTERM, lambda signum, frame: None)
            signal.set_wakeup_fd(w.fileno(), warn_on_full_buffer=False)
        except (ValueError, OSError):
            return
        self._wake = (r, w)

        def waiter():
            while True:
                b = r.recv(1)
                if b and b[0] == signal.SIGTERM:
                    self.fail(self.stage, "SIGTERM from the launcher while in this stage (another rank failed or the job was cancelled)", code=4)
        threading.Thread(target=waiter, daemon=True).start()


def init_distributed(a, guard, rank, local_rank, world):
    """-> (dist module, device).  Rendezvous on 127.0.0.1 + communicator + one small collective that also proves the ranks
    sit on `world` DISTINCT devices; every part under a deadline (`--init-timeout`, default 180 s: a fresh box's first
    `import torch` alone can take two minutes, and that happens before this)."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("NCCL_DEBUG", "WARN")          # RCCL's warnings reach stderr (stdout is redirected there)
    dev = None
    sys.stderr.write("[bench rank %d] entering init_process_group(%s), world_size %d, deadline %.0f s\n" % (rank, a.backend, world, a.init_timeout))
    sys.stderr.flush()
    with guard.deadline("init_process_group(%s), world_size %d" % (a.backend, world), a.init_timeout):
        if a.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dev = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=a.init_timeout))
        else:
            dist.init_process_group(a.backend, timeout=datetime.timedelta(seconds=a.init_timeout))
            torch.cuda.set_device(local_rank)
            dev = torch.device("cuda", local_rank)
    if dist.get_world_size() != world or (a.gpus > 1 and dist.get_world_size() != a.gpus):
        guard.fail("world size", "process group has %d ranks, --gpus %d, WORLD_SIZE %d" % (dist.get_world_size(), a.gpus, world))
    with guard.deadline("first collective (all_gather of device identities + barrier)", a.init_timeout):
        small = dev if a.backend == "nccl" else torch.device("cpu")
        pr = torch.cuda.get_device_properties(dev)
        mine = torch.tensor([torch.cuda.current_device(), getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0),
                             getattr(pr, "pci_device_id", 0)], dtype=torch.int64, device=small)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        dist.barrier()
        torch.cuda.synchronize()
    ids = [tuple(int(v) for v in e.tolist()) for e in every]
    info = {"world_size": dist.get_world_size(), "devices": [i[0] for i in ids], "pci": ["%04x:%02x:%02x" % i[1:] for i in ids],
            "distinct_devices": len(set(ids)) == world}
    if not info["distinct_devices"] and not a.one_device:
        guard.fail("device placement", "ranks share a device: %s (LOCAL_RANK not honoured?)" % (info,))
    return dist, dev, info


IMG_H, IMG_W, N_CAMS = 378, 504, 17      # LLFF 'fern' at factor 8: the image size / view count of configs[1..3]


def _kernel_table(kern, steps):
    """per timed region: launches, average time, and both floors (MFMA at the dense rate of the pipe the region runs
    on, HBM at 8 TB/s over its algorithmic bytes) with the binding one named"""
    from scnerf_amd import ops
    products = 3 if (ops.wgrad_arithmetic() == "half" and ops.mlp_arithmetic() == "resident") else 0
    out = {}
    for k, v in kern.items():
        e = {"avg_ms": v["avg_ms"], "launches_per_step": v["launches"] / steps}
        f = floors(k, v["avg_ms"], products)
        if f:
            e.update(f)
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
    cam.share_matrix_node = True                      # (one backward per step: K and E of a parameter version share a node)
    g = torch.Generator().manual_seed(100 + rank)
    kps = torch.stack([torch.randint(0, IMG_W, (n,), generator=g), torch.randint(0, IMG_H, (n,), generator=g)], -1)
    idx = torch.randint(0, N_CAMS, (n,), generator=g)
    # configs[3]: matched key points of this rank's image pair (1024 matches, a quarter of them outliers)
    i0, i1 = rank % N_CAMS, (rank + 1) % N_CAMS
    E, K = cam.get_extrinsic().detach().cpu(), cam.get_intrinsic().detach().cpu()
    k0, k1 = synth.matched_keypoints(IMG_H, IMG_W, K, E[i0], E[i1], 1024, seed=8 + rank)
    return dict(net_c=make(0), net_f=make(1), cam=cam.to(dev), matches=(k0.to(dev), k1.to(dev), i0, i1),
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


PRD_WEIGHT = 1e-4          # args.ray_dist_loss_weight of the reference's demo configuration


def learnable_camera_step(w, reducer, prd=False, prd_sync=False):
    """configs[2] (and with `prd` configs[3]): key-point rays through the learnable camera model, NDC through its
    intrinsics, render fwd + bwd down to the camera parameters, + the projected-ray-distance term of one image pair
    (+ the all-reduce over networks AND camera)"""
    import types
    from scnerf_amd.get_rays import get_rays_kps_use_camera
    from scnerf_amd.ray_dist_loss import proj_ray_dist_loss_single
    from scnerf_amd.render import render
    inv = 1.0 / (3 * w["n"])
    kw = dict(network_fn=w["net_c"], network_fine=w["net_f"], network_query_fn=w["query"], N_samples=S_C,
              N_importance=S_F, perturb=1.0, raw_noise_std=1.0, use_viewdirs=True, white_bkgd=False, near=0., far=1.)

    prd_args = types.SimpleNamespace(proj_ray_dist_threshold=5.0)
    i_map = torch.arange(N_CAMS).numpy()
    one = torch.ones((), device=w["target"].device)

    def step():
        reducer.zero()
        rays_o, rays_d = get_rays_kps_use_camera(H=IMG_H, W=IMG_W, camera_model=w["cam"],
                                                 idx_in_camera_param=w["idx"], kps_list=w["kps"])
        rgb, _, _, extras = render(H=IMG_H, W=IMG_W, chunk=1 << 15, rays=torch.stack([rays_o, rays_d]), retraw=True,
                                   camera_model=w["cam"], mode="train", **kw)
        g1 = (rgb.detach() - w["target"]) * (2 * inv)
        g0 = (extras["rgb0"].detach() - w["target"]) * (2 * inv)
        outs, grads = [rgb, extras["rgb0"]], [g1, g0]
        if prd:
            # configs[3]: + ray_dist_loss_weight x the projected-ray-distance loss of one image pair (run_nerf.py:533-596)
            k0, k1, i0, i1 = w["matches"]
            r0 = get_rays_kps_use_camera(H=IMG_H, W=IMG_W, camera_model=w["cam"], idx_in_camera_param=i0, kps_list=k0)
            r1 = get_rays_kps_use_camera(H=IMG_H, W=IMG_W, camera_model=w["cam"], idx_in_camera_param=i1, kps_list=k1)
            # (the match count stays on the device: the reference's python float is a host sync in the middle of the step, and
            #  its caller only logs it -- ray_dist_loss.py `_sync`; profiles/r06_bench_n1_config3*.json carry both)
            loss, _ = proj_ray_dist_loss_single(kps0_list=k0, kps1_list=k1, img_idx0=i0, img_idx1=i1, rays0=r0, rays1=r1,
                                                mode="train", device=k0.device, H=IMG_H, W=IMG_W, args=prd_args,
                                                camera_model=w["cam"], method="NeRF", i_map=i_map, _sync=prd_sync)
            outs.append(loss)
            grads.append(one * PRD_WEIGHT)
        torch.autograd.backward(outs, grads)
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
    # the headline step with EVERYTHING on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): the arithmetic of round 1
    from scnerf_amd import ops
    saved = (ops.mlp_arithmetic(), ops.wgrad_arithmetic())
    try:
        ops.mlp_arithmetic("fp32")
        ops.wgrad_arithmetic("fp32")
        red = FlatGradAllReduce([w["net_c"], w["net_f"]], 1)
        ms, kern = _timed(fixed_camera_step(w, red), 5, 2, sync)
        out["all_fp32_mfma_step"] = {"ms_per_step": ms, "rays_per_s": w["n"] / (ms * 1e-3),
                                     "step_tflops_over_fp32_mfma_peak": 3 * FLOP_PER_SAMPLE_FWD * (S_C + S_C + S_F) * w["n"]
                                     / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, "kernels": kern}
    finally:
        ops.mlp_arithmetic(saved[0])
        ops.wgrad_arithmetic(saved[1])
    # PSNR vs reference: 300 training steps on the procedural scene against the CPU oracle's trajectory of the same run
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import psnr_trajectory
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "psnr_oracle.json")))
        t0 = time.perf_counter()
        curve = psnr_trajectory.run_gpu(gold["steps"], 25, saved[0])
        want = {c["step"]: c["psnr"] for c in gold["curve"]}
        out["psnr_vs_reference"] = {
            "workload": "procedural scene, %d steps of %d rays x (%d + %d), Adam lr %g, identical init / batches / randoms on "
                        "both sides; PSNR of the fine render on 2048 held-out rays (tools/psnr_trajectory.py)"
                        % (gold["steps"], gold["n_rand"], gold["samples"][0], gold["samples"][1], gold["lr"]),
            "arithmetic": saved[0], "final_psnr_gpu": curve[-1]["psnr"], "final_psnr_cpu_oracle": gold["final_psnr"],
            "final_psnr_cpu_oracle_ensemble_mean_std": [gold["mean"][-1], gold["std"][-1]],
            "max_abs_psnr_difference_along_the_curve": max(abs(c["psnr"] - want[c["step"]]) for c in curve if c["step"] in want),
            "note": "single trajectories of a chaotic training run differ by 0.05-0.2 dB where the curve is steep (the oracle's "
                    "own 1e-7-perturbed ensemble: tests/golden/psnr_oracle.json); ensemble means agree within 0.1 dB "
                    "(tests/test_gpu_psnr.py, profiles/psnr_r03.json)",
            "seconds_gpu": time.perf_counter() - t0, "seconds_cpu_oracle": gold["seconds"]}
    except Exception as e:                       # (reported, never fatal to the headline line)
        out["psnr_vs_reference"] = {"error": repr(e)}
    return out


def rccl_allreduce_probe(dev, n_floats, dist_mod=None):
    """RCCL exercised on hardware in the single-GPU run too: a process group of ONE rank (backend nccl = RCCL), the
    step's collective -- one all-reduce of the flat fp32 gradient buffer (networks + camera) -- timed over 20 calls."""
    import torch.distributed as dist
    own = False
    try:
        if not dist.is_initialized():
            import socket
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
            s.close()
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
            own = True
        buf = torch.zeros(n_floats, dtype=torch.float32, device=dev)
        for _ in range(3):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        return {"ms_per_all_reduce": e0.elapsed_time(e1) / 20, "floats": n_floats, "world_size": dist.get_world_size(),
                "backend": dist.get_backend()}
    except Exception as e:
        return {"error": repr(e)}
    finally:
        if own and dist.is_initialized():
            dist.destroy_process_group()


def self_launch(n, json_out):
    """`python bench.py --gpus N` without a launcher: re-executes this command line under torch.distributed.run with N
    ranks on this node and hands rank 0's JSON line through to the real stdout.  -> exit status"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if lines:
        json_out.write(lines[-1] + "\n")
    else:                                            # (rank 0 died without a word: the launcher's status is all there is)
        json_out.write(json.dumps({"metric": METRIC, "value": None, "unit": "rays/s", "n_gpus": n, "error":
                                   "no JSON line from rank 0; torch.distributed.run exited with status %d" % r.returncode,
                                   "stage": "launch"}) + "\n")
    json_out.flush()
    bad = (not lines) or ("\"error\"" in lines[-1])
    return r.returncode if r.returncode else (1 if bad else 0)


class Telemetry:
    """Socket power and shader clock of this rank's GPU while the step loops, from the amdgpu hwmon files of its PCI device
    (power1_input in microwatts, freq1_input = sclk in Hz, power1_cap): the resident kernels run AT the socket's power cap,
    well below 2.4 GHz (profiles/r05_lab_residency.txt), and the record should say so.  A thread samples every 25 ms; nothing
    here is inside the timed region.  Every failure (no sysfs, other driver layout) yields None."""

    def __init__(self, dev):
        import glob
        self.dir = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            hits = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
            if hits and os.path.isfile(os.path.join(hits[0], "power1_input")):
                self.dir = hits[0]
        except Exception:
            self.dir = None

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as fh:
                return float(fh.read().strip())
        except Exception:
            return None

    def start(self, period=0.010):
        """samples in a thread until stop(): used AROUND THE TIMED STEPS themselves (two small sysfs reads every 10 ms in
        another thread; the launches are asynchronous and queued far ahead, the GPU never waits for the host here)"""
        import threading
        self._stop, self._power, self._clock = threading.Event(), [], []
        if self.dir is None:
            return

        def sampler():
            while not self._stop.is_set():
                pw, ck = self._read("power1_input"), self._read("freq1_input")
                if pw is not None:
                    self._power.append(pw * 1e-6)
                if ck is not None:
                    self._clock.append(ck * 1e-9)
                time.sleep(period)
        self._th = threading.Thread(target=sampler, daemon=True)
        self._th.start()

    def stop(self):
        if self.dir is None:
            return None
        self._stop.set()
        self._th.join()
        power, clock = sorted(self._power), sorted(self._clock)
        if len(power) < 4 or len(clock) < 4:
            return None
        cap = self._read("power1_cap")
        return {"socket_power_w": power[len(power) // 2], "clock_ghz": clock[len(clock) // 2],
                "power_cap_w": cap * 1e-6 if cap else None, "samples": len(power),
                "source": "amdgpu hwmon power1_input / freq1_input (sclk), median of %d samples taken every 10 ms DURING the "
                          "timed steps" % len(power)}

    def sample_while(self, fn, seconds):
        """runs fn() repeatedly for `seconds`, sampling beside it -> dict or None"""
        if self.dir is None:
            return None
        import threading
        stop, power, clock = threading.Event(), [], []

        def sampler():
            while not stop.is_set():
                pw, ck = self._read("power1_input"), self._read("freq1_input")
                if pw is not None:
                    power.append(pw * 1e-6)
                if ck is not None:
                    clock.append(ck * 1e-9)
                time.sleep(0.025)
        t_end = time.perf_counter() + seconds
        for _ in range(10):                       # (the first samples would still see the idle clock)
            fn()
        torch.cuda.synchronize()
        th = threading.Thread(target=sampler, daemon=True)
        th.start()
        steps = 0
        t0 = time.perf_counter()
        while time.perf_counter() < t_end:
            fn()
            steps += 1
            if steps % 8 == 0:
                torch.cuda.synchronize()          # (keeps the launch queue short, so the loop ends on time)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        stop.set()
        th.join()
        if not power or not clock:
            return None
        power.sort(), clock.sort()
        cap = self._read("power1_cap")
        return {"socket_power_w": power[len(power) // 2], "clock_ghz": clock[len(clock) // 2],
                "power_cap_w": cap * 1e-6 if cap else None, "samples": len(power), "ms_per_step_during": dt / max(steps, 1) * 1e3,
                "source": "amdgpu hwmon power1_input / freq1_input (sclk), median of samples taken every 25 ms over %.1f s of "
                          "back-to-back steps after the timed region" % seconds}


COMPACT_LIMIT = 4096        # bytes: the driver parses the LAST line of a bounded stdout tail


def compact_record(full, detail_path):
    """The one line the driver reads: the contract's keys, `roofline` and `cpu_baseline` trimmed to their numbers, the
    all-fp32-MFMA step beside the headline, and where the full record (per-kernel tables, other configurations, prose)
    was written.  Stays under COMPACT_LIMIT bytes (tests/test_bench_line.py)."""
    def pick(d, keys):
        return None if d is None else {k: d[k] for k in keys if k in d}

    def r4(x):
        return float("%.5g" % x) if isinstance(x, float) else x

    def rounded(d):
        return None if d is None else {k: r4(v) for k, v in d.items()}
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data") if k in full}
    line["config"] = pick(full.get("config"), ("workload", "baseline_config", "rays_per_gpu", "parallelism"))
    line["roofline"] = rounded(pick(full.get("roofline"), (
        "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
        "traffic_over_survey_algorithmic", "avg_launch_ms", "frac_mfma", "frac_hbm", "algorithmic_bytes_per_launch",
        "survey_algorithmic_bytes_per_launch", "flop_per_launch", "step_traffic_bytes", "step_traffic_over_survey_algorithmic",
        "clock_ghz", "socket_power_w", "power_cap_w", "mfma_busy", "clock_ghz_under_counters", "frac_at_measured_clock",
        "traffic_measured_live")))
    line["cpu_baseline"] = rounded(pick(full.get("cpu_baseline"), (
        "value", "unit", "cores", "kind", "sample", "best_ms", "median_ms", "value_median", "os_cpu_count",
        "rays_per_s_1024_rays_anomaly_on_as_shipped")))
    if "speedup_vs_cpu_baseline" in full:
        line["speedup_vs_cpu_baseline"] = r4(full["speedup_vs_cpu_baseline"])
    fp32 = (full.get("extras") or {}).get("all_fp32_mfma_step")
    if fp32 and "ms_per_step" in fp32:
        line["all_fp32_mfma_step"] = {"ms_per_step": r4(fp32["ms_per_step"]), "rays_per_s": r4(fp32["rays_per_s"]),
                                      "frac_of_157.3_TF": r4(fp32["step_tflops_over_fp32_mfma_peak"])}
    for k in ("ms_per_step_events_off", "step_tflops", "per_rank_ms_per_step"):
        if k in full:
            line[k] = [r4(x) for x in full[k]] if isinstance(full[k], list) else r4(full[k])
    if full.get("all_reduce_alone"):
        line["all_reduce_alone"] = rounded(pick(full["all_reduce_alone"], ("ms_per_all_reduce", "floats", "world_size", "backend", "error")))
    if full.get("ranks"):
        line["ranks"] = pick(full["ranks"], ("world_size", "devices", "distinct_devices"))
    if "infer_rays_per_s" in full:
        line["infer_rays_per_s"] = r4(full["infer_rays_per_s"])
    line["detail"] = detail_path
    text = json.dumps(line)
    if len(text) >= COMPACT_LIMIT:                  # (cannot happen with the fields above; never emit an unparseable tail)
        for k in ("all_reduce_alone", "per_rank_ms_per_step", "all_fp32_mfma_step"):
            line.pop(k, None)
        if line.get("cpu_baseline"):
            line["cpu_baseline"]["sample"] = str(line["cpu_baseline"].get("sample"))[:200]
        if line.get("config"):
            line["config"]["workload"] = str(line["config"].get("workload"))[:200]
        text = json.dumps(line)
    return text


def latest_pmc_traffic():
    """the newest profiles/pmc_traffic_r*.json whose source hash matches the kernel sources (None: stale or absent)"""
    import glob
    want = csrc_sha16()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_traffic_r*.json")), reverse=True):
        rec = json.load(open(path))
        if rec.get("_csrc_sha16") == want:
            return rec, os.path.relpath(path, ROOT), None
        stale = stale or "%s was taken at other kernel sources (%s, now %s): refused as stale" % (
            os.path.relpath(path, ROOT), rec.get("_csrc_sha16"), want)
    return None, None, stale


def live_pmc_traffic(dominant_region, rays, steps=3, timeout=150):
    """HBM traffic measured in THIS run, as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes: two separate passes of
    `rocprofv3 --kernel-trace --pmc <ONE counter>` (FETCH_SIZE, WRITE_SIZE; nothing else traced) over `steps` plain steps of
    the headline workload in a child process (tools/profile_steps.py pmc), bytes = WRITE_SIZE KB x 1024 + 2 x FETCH_SIZE KB x
    1024 (gfx950 tallies wide coalesced reads at half their size).  -> (dict | None, note): per launch of the dominant kernel
    and per step (every dispatch of the child / steps: a few MB of set-up kernels are in the sum)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.isfile(exe):
        return None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from pmc_summary import REGIONS
    except Exception as e:
        return None, "tools/pmc_summary.py: %r" % (e,)
    import re
    anyp = lambda name: re.sub(r"/P=\d+", "/P=*", name)         # (the table names its regions at the headline's sample counts)
    patterns = [pat for pat, region in REGIONS.items() if anyp(region) == anyp(dominant_region)]
    tmp = tempfile.mkdtemp(prefix="scnerf_pmc_", dir="/tmp")
    total, per_launch = {}, {}
    t0 = time.perf_counter()
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, c)
            env = dict(os.environ, TMPDIR="/tmp", PMC_STEPS=str(steps), PMC_RAYS=str(rays))
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            try:
                r = subprocess.run([exe, "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", d, "-o", c, "--",
                                    sys.executable, os.path.join(ROOT, "tools", "profile_steps.py"), "pmc"],
                                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=timeout)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s did not finish within %d s" % (c, timeout)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, "rocprofv3 --pmc %s left no counter file (status %d): %s" % (c, r.returncode, (r.stderr or "")[-300:])
            tot, mine = 0.0, []
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != c:
                    continue
                v = float(row["Counter_Value"])
                tot += v
                if any(pat in row["Kernel_Name"] for pat in patterns):
                    mine.append((int(row.get("Grid_Size") or 0), v))
            total[c] = tot
            # (a kernel that serves both passes under one name -- the data-gradient kernel -- is launched at two sizes per step:
            #  the dominant REGION is the larger one, the fine pass)
            big = [v for g, v in mine if g == max(g_ for g_, _ in mine)] if mine else []
            per_launch[c] = sum(big) / len(big) if big else None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {"bytes_per_step": int((total["WRITE_SIZE"] * 1024 + 2 * total["FETCH_SIZE"] * 1024) / steps),
           "bytes_per_launch": (int(per_launch["WRITE_SIZE"] * 1024 + 2 * per_launch["FETCH_SIZE"] * 1024)
                                if per_launch["FETCH_SIZE"] is not None and per_launch["WRITE_SIZE"] is not None else None),
           "seconds": time.perf_counter() - t0}
    return out, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two separate passes, one counter "
                 "each) over %d plain steps of the headline workload in a child process (tools/profile_steps.py pmc); bytes = "
                 "WRITE_SIZE KB x 1024 + 2 x FETCH_SIZE KB x 1024" % steps)


def main():
    # stdout carries exactly ONE line, the JSON record: everything else that writes to file descriptor 1 (RCCL prints a
    # version banner there when its first communicator comes up, progress bars, ...) is sent to stderr
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=4096, help="rays of the CPU reference leg (BASELINE.md section 2: 4096)")
    ap.add_argument("--detail", default=None, help="where the full record goes (default profiles/bench_detail_n<N>.json)")
    ap.add_argument("--config", type=int, choices=(1, 2, 3, 4), default=1,
                    help="BASELINE.json's configs[k], the SAME at every --gpus N: 1 fixed camera (default, the headline); "
                         "2 learnable camera model; 3 camera model + projected-ray-distance term every step; 4 NeRF++")
    ap.add_argument("--camera", action="store_true", help="(= --config 2)")
    ap.add_argument("--mlp-arithmetic", choices=("resident", "fp32"), default=None,
                    help="forward and data-gradient chain: 'resident' (default): the whole network as one launch each on "
                         "three fp16 products with register-resident activations; 'fp32': the fused fp32-MFMA kernels "
                         "(the numerical yardstick)")
    ap.add_argument("--wgrad-arithmetic", choices=("half", "fp32"), default=None,
                    help="weight-gradient GEMMs: three fp16 products with a scale per operand and workgroup chunk (default; "
                         "needs the resident kernels' chunk maxima), or the exact-fp32 MFMA")
    ap.add_argument("--backend", default=os.environ.get("SCNERF_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend (nccl = RCCL; gloo for a functional check of N ranks on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="all ranks on cuda:0 (functional check, with --backend gloo)")
    ap.add_argument("--telemetry-seconds", type=float, default=1.5,
                    help="N = 1: socket power and shader clock sampled over this many seconds of steps after the timed region (0: off)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="N = 1, --config 1: do not run the two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; ~40 s) that measure "
                         "`roofline.traffic` in this run; the newest committed summary at the same kernel sources is quoted instead")
    ap.add_argument("--prd-sync", action="store_true",
                    help="--config 3: proj_ray_dist_loss_single returns the match count as a python float, as the reference does "
                         "(a host sync per step); default: it stays on the device")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --rays is the TOTAL batch, split evenly over the ranks (SURVEY section 8d, C4's second "
                         "figure); default: weak scaling, --rays per GPU")
    ap.add_argument("--init-timeout", type=float, default=float(os.environ.get("SCNERF_BENCH_INIT_TIMEOUT", "180")),
                    help="N > 1: seconds the rendezvous + communicator start-up, and then the first collective, may take before "
                         "rank 0 prints an error line and every rank exits non-zero")
    a = ap.parse_args()
    scale = float(os.environ.get("SCNERF_BENCH_TIMEOUT_SCALE", "1"))
    a.init_timeout *= scale

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: one process per GPU through torch.distributed.run (what the driver's own command
        # line does), rendezvous on 127.0.0.1; rank 0's JSON line is the only thing on the children's stdout
        raise SystemExit(self_launch(a.gpus, json_out))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    guard = Guard(json_out, rank, world, a)
    guard.install_sigterm()
    try:
        run(a, json_out, guard, rank, world, scale)
    except SystemExit:
        raise
    except BaseException as e:                       # (every failure leaves a parseable line and a non-zero status)
        import traceback
        traceback.print_exc()
        guard.fail(guard.stage, "%s: %s" % (type(e).__name__, e), code=1)


def run(a, json_out, guard, rank, world, scale):
    local_rank = 0 if a.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        guard.fail("launch", "--gpus %d under a launcher that started %d ranks" % (a.gpus, world), code=2)
    dist, ranks_info = None, None
    if world > 1:
        dist, dev, ranks_info = init_distributed(a, guard, rank, local_rank, world)
    else:
        guard.stage = "device"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    guard.stage = "build the workload"

    from scnerf_amd import ops
    from scnerf_amd.parallel import FlatGradAllReduce
    ops.check_layout()
    if a.wgrad_arithmetic:
        ops.wgrad_arithmetic(a.wgrad_arithmetic)
    if a.mlp_arithmetic:
        ops.mlp_arithmetic(a.mlp_arithmetic)
    n = a.rays // world if a.strong else a.rays
    if a.strong and (n < 1 or a.rays % world):
        guard.fail("arguments", "--strong needs --rays divisible by the %d ranks" % world, code=2)
    cfg = 2 if (a.camera and a.config == 1) else a.config
    step_flop = 3 * FLOP_PER_SAMPLE_FWD * (S_C + S_C + S_F) * n
    if cfg == 4:
        from tools import bench_nerfpp
        if a.rays == 4096 and not a.strong:
            n = 2048                                       # the NeRF++ step's batch (two levels x two networks)
        npp_step, nets, step_flop = bench_nerfpp.build(n, seed_offset=rank, device=dev)
        reducer = FlatGradAllReduce(nets, world)
        w = None

        def step():
            reducer.zero()
            npp_step(zero=False)
            reducer.all_reduce()
    else:
        w = build_world(dev, rank, n)
        if cfg == 1:
            reducer = FlatGradAllReduce([w["net_c"], w["net_f"]], world)
            step = fixed_camera_step(w, reducer)
        else:
            reducer = FlatGradAllReduce([w["net_c"], w["net_f"], w["cam"]], world)
            step = learnable_camera_step(w, reducer, prd=(cfg == 3), prd_sync=a.prd_sync)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # (deadlines: 60 s + 2 s per step is two orders of magnitude above a healthy run; a rank stuck in the collective ends here)
    with guard.deadline("warm-up steps (the first all-reduce of the flat gradient buffer is in here)", (60 + 2 * a.warmup) * scale):
        for _ in range(a.warmup):
            step()
        sync()
    live = Telemetry(dev) if (rank == 0 and a.telemetry_seconds > 0) else None
    with guard.deadline("timed steps", (60 + 2 * a.steps) * scale):
        ops.PROFILE.reset(enabled=True)
        if live:
            live.start()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        sync()
        dt = time.perf_counter() - t0
        in_loop = live.stop() if live else None
        ops.PROFILE.enabled = False
    per_rank_ms, allreduce_info = None, None
    if world > 1:
        small = dev if a.backend == "nccl" else torch.device("cpu")      # (gloo gathers host tensors)
        mine = torch.tensor([dt / a.steps * 1e3], device=small, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(x.item()) for x in every]
        tt = torch.tensor([dt], device=small, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        with guard.deadline("the collective on its own", 120 * scale):
            allreduce_info = rccl_allreduce_probe(dev, int(reducer.flat.numel()))       # the step's collective on its own
    ms = dt / a.steps * 1e3
    guard.stage = "after the timed region"

    # the same K steps without the event pairs (outside the reported figure)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    ms_off = (time.perf_counter() - t0) / a.steps * 1e3
    # clock / power: the samples taken during the timed steps when there are enough of them (>= 4: ~40 ms of steps), else
    # a loop of its own after the timed region (labelled as such)
    telemetry = None
    if rank == 0 and world == 1 and a.telemetry_seconds > 0:
        # socket power: amdgpu's power1_input is a moving average over about a second -- 0.24 s of timed steps only see it
        # ramp --, so it is read over a loop of its own; the shader clock responds at once and is the timed steps' own
        telemetry = Telemetry(dev).sample_while(step, a.telemetry_seconds)
        if telemetry and in_loop:
            telemetry["clock_ghz_after_the_timed_steps"] = telemetry["clock_ghz"]
            telemetry["clock_ghz"] = in_loop["clock_ghz"]
            telemetry["source"] = ("amdgpu hwmon: freq1_input (sclk) = median of %d samples taken every 10 ms DURING the timed steps; "
                                   "power1_input (a ~1 s moving average) over %.1f s of back-to-back steps after them"
                                   % (in_loop["samples"], a.telemetry_seconds))
    elif in_loop:
        telemetry = in_loop

    if rank == 0:
        kern = ops.PROFILE.summary()
        table = _kernel_table(kern, a.steps)
        # dominant kernel: the single launch with the largest time per step
        single = {k: v for k, v in kern.items() if not v["group"] and region_model(k)}
        dom = max(single, key=lambda k: single[k]["total_ms"]) if single else None
        roof = None
        if dom:
            k, f = kern[dom], table[dom]
            hbm = f["bound"] == "hbm"
            products = region_model(dom)["products"]
            peak = PEAK_HBM_GBS if hbm else (PEAK_16BIT_MFMA_TFLOPS / products if products else PEAK_F32_MFMA_TFLOPS)
            ach = f["gbytes_per_s"] if hbm else f["tflops"]
            roof = {"bound": f["bound"], "kernel": dom, "achieved": ach, "peak": peak, "unit": "GB/s" if hbm else "TFLOP/s",
                    "frac": ach / peak, "traffic": None, "frac_mfma": f["frac_mfma"], "frac_hbm": f["frac_hbm"],
                    "floor_mfma_ms": f["floor_mfma_ms"], "floor_hbm_ms": f["floor_hbm_ms"],
                    "frac_hbm_of_measured_copy_rate": f["frac_hbm"] * PEAK_HBM_GBS / MEASURED_HBM_GBS,
                    "avg_launch_ms": k["avg_ms"], "launches_per_step": k["launches"] / a.steps,
                    "flop_per_launch": f["flop_per_launch"], "algorithmic_bytes_per_launch": f["algorithmic_bytes_per_launch"],
                    "pipe": f["pipe"],
                    "peak_note": ("TFLOP/s are algorithmic fp32 products per second; the MFMA peak is the dense 16-bit rate "
                                  "(2500) over the 3 partial products per fp32 product (two-way fp16 cut), or the fp32 MFMA's "
                                  "157.3; the HBM peak is the 8 TB/s spec over the ALGORITHMIC bytes "
                                  "of the launch.  `bound` names the larger of the two floors, `frac` = that floor / the "
                                  "measured time; frac_mfma and frac_hbm are both given"),
                    "measured": ("HIP events on the launch stream around each launch of this kernel inside the timed region "
                                 "(ops.PROFILE; the eight 256 x 256 weight-gradient GEMMs: events recorded by the C entry point "
                                 "around its one launch, scnerf_wgrad_profile_events)")}
            # SURVEY section 8(d)'s algorithmic bytes: what crosses HBM if nothing but the path's inputs and outputs does
            # (the point in, raw out for a network launch; per step 5 KB per ray): the activation / gradient workspaces the
            # weight-gradient GEMMs read are the design's own traffic, reported against it
            md = region_model(dom)
            pd = 4 if "/pd4" in dom else 3
            import re as _re
            P_dom = int(_re.search(r"/P=(\d+)", dom).group(1))
            roof["survey_algorithmic_bytes_per_launch"] = (4 * pd + 16) * P_dom
            rec, path, stale = latest_pmc_traffic()
            live, live_note = (None, "not requested")
            if world == 1 and cfg == 1 and not a.no_pmc:
                guard.stage = "rocprofv3 counter passes (HBM traffic of this run)"
                try:
                    live, live_note = live_pmc_traffic(dom, n)
                except Exception as e:                # (never fatal to the line)
                    live, live_note = None, "failed: %r" % (e,)
            if live and live.get("bytes_per_launch"):
                # measured in THIS run; the committed summary stays the source of the matrix-pipe utilisation below
                roof["traffic"] = live["bytes_per_launch"]
                roof["traffic_source"] = live_note
                roof["traffic_over_algorithmic"] = roof["traffic"] / f["algorithmic_bytes_per_launch"]
                roof["traffic_over_survey_algorithmic"] = roof["traffic"] / roof["survey_algorithmic_bytes_per_launch"]
                roof["step_traffic_bytes"] = live["bytes_per_step"]
                roof["step_traffic_over_survey_algorithmic"] = live["bytes_per_step"] / (5000.0 * n)
                roof["traffic_measured_live"] = True
                roof["traffic_pass_seconds"] = live["seconds"]
                if rec and rec.get("bytes_per_launch", {}).get(dom):
                    roof["traffic_committed_summary"] = rec["bytes_per_launch"][dom]
            elif rec and cfg == 1 and n == 4096:        # (the committed summary was taken on the headline workload: quoted for it alone)
                roof["traffic_measured_live"] = False
                roof["traffic_live_note"] = live_note
                roof["traffic"] = rec.get("bytes_per_launch", {}).get(dom)
                roof["traffic_source"] = "%s: %s" % (path, rec.get("_source"))
                if roof["traffic"]:
                    roof["traffic_over_algorithmic"] = roof["traffic"] / f["algorithmic_bytes_per_launch"]
                    roof["traffic_over_survey_algorithmic"] = roof["traffic"] / roof["survey_algorithmic_bytes_per_launch"]
                if rec.get("bytes_per_step"):
                    roof["step_traffic_bytes"] = rec["bytes_per_step"]
                    roof["step_traffic_over_survey_algorithmic"] = rec["bytes_per_step"] / (5000.0 * n)
            else:
                roof["traffic_source"] = live_note if (cfg != 1 or n != 4096 or world > 1) else (stale or "no profiles/pmc_traffic_r*.json")
            # the power-capped ceiling: the chip's 16-bit MFMA peak is quoted at 2.4 GHz; under these kernels the socket sits at
            # its power cap and clocks lower, so the same kernel is also priced against the peak AT THE CLOCK IT RAN AT
            if rec and cfg == 1 and n == 4096 and rec.get("mfma_busy", {}).get(dom) is not None:
                roof["mfma_busy"] = rec["mfma_busy"][dom]
                roof["clock_ghz_under_counters"] = rec.get("clock_ghz", {}).get(dom)
            if telemetry:
                roof["clock_ghz"] = telemetry["clock_ghz"]
                roof["socket_power_w"] = telemetry["socket_power_w"]
                roof["power_cap_w"] = telemetry["power_cap_w"]
                # (an ESTIMATE beside `frac`, never instead of it: frac_mfma rescaled from 2.4 GHz to the median shader clock
                #  sampled over the same timed steps -- or, on runs too short for four samples, over a loop after them)
                roof["frac_at_measured_clock"] = roof["frac_mfma"] * 2.4 / telemetry["clock_ghz"] if not hbm else None
                roof["telemetry"] = telemetry
        workload = {
            1: "configs[1]: %d rays x (64 coarse + 128 fine), coarse+fine NeRF (D=8, W=256), fwd+bwd of render_rays per GPU, "
               "perturb=1, raw_noise_std=1; precomputed rays of a fixed camera" % n,
            2: "configs[2]: %d rays x (64 + 128) per GPU, coarse+fine NeRF, rays from the learnable camera model (%d views, "
               "%dx%d: intrinsics, extrinsics, ray-o / ray-d noise), fwd+bwd down to the camera parameters" % (n, N_CAMS, IMG_H, IMG_W),
            3: "configs[3]: %d rays x (64 + 128) per GPU, coarse+fine NeRF, rays from the learnable camera model (%d views, "
               "%dx%d) + the projected-ray-distance loss of one image pair (1024 matches) in every step, fwd+bwd down to the "
               "camera parameters; match count %s" % (n, N_CAMS, IMG_H, IMG_W, "read on the host per step as the reference does"
                                                      if a.prd_sync else "left on the device"),
            4: "configs[4]: NeRF++ step, %d rays per GPU, two cascade levels (64, then 64+128 samples) x (foreground + "
               "background network), fwd+bwd" % n}[cfg]
        collective = ("no collective at N = 1" if world == 1 else
                      "1 %s all-reduce/step of %d floats" % ("RCCL" if a.backend == "nccl" else a.backend, int(reducer.flat.numel())))
        out = {
            "metric": metric_for(cfg), "value": n * world / (ms * 1e-3),
            "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "ms_per_step_events_off": ms_off,
            "higher_is_better": True, "scaling": "strong" if a.strong else "weak", "vs_baseline": None,
            "dtype": {"resident": "f32 (3 x fp16 MFMA products per product, fp32 accumulate)", "fp32": "f32 (fp32 MFMA)"}[ops.mlp_arithmetic()],
            "arithmetic": {
                "forward and data gradients": {
                    "resident": "the whole network as ONE launch per pass (csrc/mlp_h3.h): every operand scaled by a power of "
                                "two (per sample from the row / column 1-norm bound x the measured input maximum; per layer for "
                                "the weights) and cut into 2 fp16 numbers, 3 partial products on v_mfma_f32_32x32x16_f16, fp32 "
                                "accumulate; activations register-resident from layer to layer; per-layer error vs fp64 = the "
                                "fp32 MFMA's (profiles/parity_r05.json resident_layer_arithmetic_*)",
                    "fp32": "fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32 (fused kernels)"}[ops.mlp_arithmetic()],
                "256x256 weight gradients": {
                    "half": "fp32 operands scaled by one power of two per operand and workgroup chunk (the chunk maxima come "
                            "from the resident kernels) and cut into 2 fp16 numbers, 3 partial products on "
                            "v_mfma_f32_32x32x16_f16, fp32 accumulate (csrc/wgrad256_half.h); error vs fp64 = the fp32 MFMA "
                            "kernel's (profiles/parity_r05.json wgrad256_arithmetic_*)",
                    "fp32": "fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32"}[
                        "half" if (ops.wgrad_arithmetic() == "half" and ops.mlp_arithmetic() == "resident") else "fp32"],
                "narrow weight gradients": (
                    "256 x 64 (encoded point, twice) and 128 x (256 + 32) (views layer): three fp16 products as the 256 x 256 "
                    "ones, scales from the same chunk maxima (csrc/wgrad_half_narrow.h); rgb rows and alpha_linear's "
                    "vector-matrix product: fp32 VALU at the HBM rate" if ops.wgrad_arithmetic() == "half" and ops.mlp_arithmetic() == "resident"
                    else "fp32 in, fp32 accumulate: v_mfma_f32_32x32x2_f32")},
            "timed_region": ("exactly the product path: every region of `kernels` is ONE C call of the host layer between "
                             "two HIP events on the launch stream (no piecewise re-issue); ms_per_step_events_off is the same "
                             "loop without the events"),
            "data": "synthetic",
            "config": {"workload": workload, "baseline_config": cfg, "rays_per_gpu": n,
                       "parallelism": "ray-parallel x%d, %s" % (world, collective),
                       "flat_gradient_floats": int(reducer.flat.numel())},
            "roofline": roof,
            "kernels": table,
            "step_flop_algorithmic": step_flop,
            "step_tflops": step_flop / (ms * 1e-3) / 1e12,
        }
        if per_rank_ms is not None:
            out["per_rank_ms_per_step"] = per_rank_ms
            out["all_reduce_alone"] = allreduce_info
            out["ranks"] = ranks_info
        if world == 1 and not a.no_extras and cfg == 1:
            guard.stage = "extras (other configurations, inference, PSNR)"
            out["extras"] = extras_single_gpu(w, dev, sync)
            # SURVEY section 8(d)'s secondary metric: forward-only rays/s of full-image inference (render_path, 378 x 504,
            # 64 + 128 samples, incl. ray generation and the D2H copy of the finished images)
            inf = out["extras"].get("full_image_inference") or {}
            if inf.get("value"):
                out["infer_rays_per_s"] = inf["value"]
            if a.backend == "nccl":
                out["extras"]["rccl_allreduce_single_rank"] = rccl_allreduce_probe(dev, 1202945)
        # the CPU leg IS configs[1]'s render_rays: only the headline configuration is compared with it
        if world == 1 and not a.no_cpu and cfg == 1:
            guard.stage = "cpu_baseline"
            out["cpu_baseline"] = cpu_baseline(a.cpu_rays)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        # the full record goes to a file; stdout carries the compact line only (a 26 KB line once outgrew the driver's tail)
        detail = a.detail or os.path.join("profiles", "bench_detail_n%d.json" % world)
        try:
            os.makedirs(os.path.dirname(os.path.join(ROOT, detail)) or ".", exist_ok=True)
            with open(os.path.join(ROOT, detail), "w") as fh:
                json.dump(out, fh, indent=1)
        except OSError as e:
            detail = "not written: %r" % (e,)
        with guard._lock:
            guard.done = True                       # (from here on a SIGTERM must not add a second line)
            json_out.write(compact_record(out, detail) + "\n")
            json_out.flush()
    if world > 1:
        with guard.deadline("destroy_process_group", 60 * scale):
            dist.barrier()
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
