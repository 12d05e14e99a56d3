"""LAB (GPU): WHOSE kernels, running beside it from another process, make the camera backward kernel's per-ray values move?
Process 0 (victim) calls the dump kernel of tools/ubench/camera_bwd_lab.hip (alone: no memsets, no LDS stage, no finish kernel) on
fixed inputs in a tight loop and compares every dump with the first; process 1 (neighbour) cycles through modes, a few seconds each --
idle, ATen copies, a bf16 matmul, each of this library's heavy kernels on its own, the whole training step -- and publishes the mode
in shared memory.  Events are counted per mode.    python tools/guest_probe.py [SECONDS_PER_MODE] [ROUNDS]"""
import ctypes, os, sys, time
import numpy as np
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
MODES = ["idle", "aten copy", "bf16 matmul", "resident fwd", "resident dgrad", "fp32 fwd", "fp32 dgrad", "wgrad", "samplers+composite", "camera kernels", "whole step"]


def step_fn(dev):
    from tests.parallel_nerf_worker import build, batch, H, W
    from scnerf_amd.get_rays import get_rays_kps_use_camera
    from scnerf_amd.render import render
    net_c, net_f, cam, query = build(dev)
    n, sc, sf = 1025, 64, 128
    kps, idx, target, rnd = batch(n, sc, sf, dev)

    def step(keep=False):
        for m in (net_c, net_f, cam):
            for p in m.parameters(): p.grad = None
        rays_o, rays_d = get_rays_kps_use_camera(H=H, W=W, camera_model=cam, idx_in_camera_param=idx, kps_list=kps)
        if keep:
            rays_o.retain_grad(); rays_d.retain_grad()
        rgb, disp, acc, extras = render(H=H, W=W, chunk=1 << 15, rays=torch.stack([rays_o, rays_d]), retraw=True, camera_model=cam, mode="train",
                                        network_fn=net_c, network_fine=net_f, network_query_fn=query, N_samples=sc, N_importance=sf, perturb=1.0,
                                        raw_noise_std=1.0, use_viewdirs=True, white_bkgd=False, near=0., far=1., _randoms=rnd)
        loss = torch.mean((rgb - target) ** 2) + torch.mean((extras["rgb0"] - target) ** 2)
        loss.backward()
        return rays_o, rays_d
    return step, cam


def victim(mode, stop, counts, calls):
    from scnerf_amd import ops, _capi
    dev = torch.device("cuda:0")
    lab = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libcamera_bwd_lab_v0.so"))
    lab.camera_bwd_lab.argtypes = _capi.load().scnerf_camera_rays_bwd.argtypes
    lab.camera_bwd_lab.restype = ctypes.c_int
    step, cam = step_fn(dev)
    rays_o, rays_d = step(keep=True)
    c = rays_d.grad_fn.cam
    nr = int(c["n"])
    g_o, g_d = rays_o.grad.contiguous(), rays_d.grad.contiguous()
    C = int(c["extr_init"].shape[0])
    d_in = torch.empty(4, device=dev); d_ex = torch.empty((C, 9), device=dev)
    d_go = torch.empty_like(c["grid_o"]); d_gd = torch.empty_like(c["grid_d"])
    ws = torch.zeros(4 + 12 * C, device=dev); dump = torch.empty((nr, 32), device=dev)
    common = list(ops._cam_common(c)); common[4] = 7            # bare: the dump kernel alone
    P = ops._p
    ref = None
    mode.value = -2                                              # victim ready
    while not stop.value:
        m0 = mode.value
        st = lab.camera_bwd_lab(*common, P(g_o), P(g_d), P(d_in), P(d_ex), P(d_go), P(d_gd), P(ws), P(dump), nr, ops._stream())
        assert st == 0
        cur = dump.cpu().numpy().view(np.uint32)[:, :30]
        if ref is None:
            ref = cur; continue
        if m0 >= 0 and mode.value == m0:
            calls[m0] += 1
            if (cur != ref).any(): counts[m0] += 1


def neighbour(mode, stop, seconds, rounds, lib=None, only=None):
    if lib:
        os.environ["SCNERF_HIP_LIB"] = lib                      # an ablation build of the resident kernels (tools/ablate_h3.sh)
    from scnerf_amd import ops, synthetic as synth, mlp_layout as ML
    dev = torch.device("cuda:0")
    lay = ML.layout(3)
    p = synth.network_params(seed=1)
    flat = torch.cat([p[name].reshape(-1) for name, _ in lay.param_shapes]).float().to(dev)
    wpk, wbk, rw = ops.pack_weights(flat, "fwd"), ops.pack_weights(flat, "bwd"), ops.pack_resident(flat)
    rays_n, spr = 1024, 192
    Pn = rays_n * spr
    g = torch.Generator().manual_seed(5)
    pts = (torch.rand(Pn, 3, generator=g) * 3 - 1.5).to(dev)
    vd = torch.randn(rays_n, 3, generator=g); vd = (vd / vd.norm(dim=-1, keepdim=True)).to(dev)
    save = ops.save_workspace(Pn, dev)
    d_raw = (torch.randn(Pn, 4, generator=g) * 1e-3).to(dev)
    ops.mlp_fwd(pts, vd, spr, wpk, save)
    grads, _, _ = ops.mlp_bwd(d_raw, pts, vd, spr, wbk, save)
    big = torch.randn(64 << 20, device=dev)
    A = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    step, cam = step_fn(dev)
    # the small kernels of one render pass: sampler, compositing
    from tests.parallel_nerf_worker import H, W
    rays = torch.rand(rays_n, 11, device=dev)
    rays[:, 6] = 0.; rays[:, 7] = 1.
    t_vals = torch.linspace(0, 1, 64, device=dev)
    raw = torch.randn(rays_n, 64, 4, device=dev)

    def small():
        z, pts_c = ops.coarse_sample(rays, t_vals, None, False)
        ops.composite_fwd(raw, z, rays, None, False)

    def cams():
        from scnerf_amd.get_rays import get_rays_kps_use_camera
        from tests.parallel_nerf_worker import batch
        kps, idx, _, _ = batch(1025, 64, 128, dev)
        ro, rd = get_rays_kps_use_camera(H=H, W=W, camera_model=cam, idx_in_camera_param=idx, kps_list=kps)
        (ro.sum() + rd.sum()).backward()
    fns = {"idle": lambda: time.sleep(0.01), "aten copy": lambda: big.clone(), "bf16 matmul": lambda: A @ A,
           "resident fwd": lambda: ops.mlp_fwd_resident(pts, vd, spr, wpk, rw, save), "resident dgrad": lambda: ops.mlp_bwd_resident(d_raw, pts, vd, spr, wbk, rw, save),
           "fp32 fwd": lambda: ops.mlp_fwd(pts, vd, spr, wpk, save), "fp32 dgrad": lambda: ops.mlp_bwd(d_raw, pts, vd, spr, wbk, save),
           "wgrad": lambda: ops.nerf_wgrad(save, grads, d_raw, Pn), "samplers+composite": small, "camera kernels": cams, "whole step": step}
    for fn in fns.values(): fn()
    torch.cuda.synchronize()
    while mode.value != -2: time.sleep(0.05)
    for r in range(rounds):
        for i, name in enumerate(MODES):
            if only and name not in only: continue
            torch.cuda.synchronize()
            mode.value = i
            t0 = time.time()
            while time.time() - t0 < seconds:
                for _ in range(4): fns[name]()
                torch.cuda.synchronize()
            mode.value = -1
    stop.value = 1


def run(seconds, rounds, lib=None, only=None):
    ctx = mp.get_context("spawn")
    mode, stop = ctx.Value("i", -3), ctx.Value("i", 0)
    counts, calls = ctx.Array("i", len(MODES)), ctx.Array("i", len(MODES))
    ps = [ctx.Process(target=victim, args=(mode, stop, counts, calls)), ctx.Process(target=neighbour, args=(mode, stop, seconds, rounds, lib, only))]
    for p in ps: p.start()
    for p in ps: p.join()
    for i, name in enumerate(MODES):
        if only is None or name in only:
            print("%-28s %-34s %12d   %d" % (name, lib or "product", calls[i], counts[i]), flush=True)


def run_external(seconds, label):
    """the victim alone, counted for `seconds` once it is ready; the neighbour is whatever else the shell started on the GPU"""
    ctx = mp.get_context("spawn")
    mode, stop = ctx.Value("i", -3), ctx.Value("i", 0)
    counts, calls = ctx.Array("i", len(MODES)), ctx.Array("i", len(MODES))
    p = ctx.Process(target=victim, args=(mode, stop, counts, calls))
    p.start()
    while mode.value != -2: time.sleep(0.05)
    mode.value = 0
    time.sleep(seconds)
    stop.value = 1
    p.join()
    print("%-28s %-34s %12d   %d" % ("external", label, calls[0], counts[0]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--external":       # python tools/guest_probe.py --external SECONDS LABEL
        run_external(float(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "?")
        sys.exit(0)
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print("neighbour mode               neighbour's library                victim calls   calls whose per-ray dump moved")
    if len(sys.argv) > 3:                                       # python tools/guest_probe.py 3 1 "resident dgrad,resident fwd" lib1.so lib2.so ...
        only = sys.argv[3].split(",")
        for lib in sys.argv[4:] or [None]:
            run(seconds, rounds, None if lib == "product" else lib, only)
    else:
        run(seconds, rounds)
