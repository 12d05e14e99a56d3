"""LAB (GPU): the camera backward kernel with a per-ray dump (tools/ubench/camera_bwd_lab.hip), called again and again on identical
inputs while the process (and, with two processes, its neighbour on the same GPU) repeats the parallel worker's step.  Dumps are
compared bit for bit with the first call's: which per-ray value moves, in which rays / lanes, on which CU -- or whether only the
sums move.    python tools/flaky_probe5.py ITER K NPROCS"""
import ctypes, os, sys
import numpy as np
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
VARIANTS = ("v4",)      # v4: columns 0..5 are tap (y0,x0) x, y, z as loaded, tap (y0,x1).y, wx0, the first blend y
FLAGS = {"nolds": 1, "nomemset": 2, "nofinish": 4, "bare": 7}       # tools/ubench/build_camera_bwd_lab.sh: product source / taps as one-word loads / warm-up touch of the grid
COLS = ["fx", "fy", "cx", "cy", "x", "y", "dirs.x", "dirs.y", "dirs.z", "Rx.x", "Rx.y", "Rx.z", "Ry.x", "Ry.y", "Ry.z", "rd.x", "rd.y", "rd.z", "nrm",
        "gr.x", "gr.y", "gr.z", "gdx", "gdy", "gk0", "gk1", "g_in.x", "g_in.y", "g_in.z", "cam", "wave_gk0", "hw"]


def hw_str(bits):
    b = int(bits)
    return "xcc%d se%d cu%d simd%d" % (b >> 28, (b >> 13) & 7, (b >> 8) & 15, (b >> 4) & 3)


def loop(rank, iters, K, world):
    from tests.parallel_nerf_worker import build, batch, H, W
    from scnerf_amd.get_rays import get_rays_kps_use_camera
    from scnerf_amd.render import render
    from scnerf_amd import ops
    from scnerf_amd import _capi
    labs = {}
    for v in VARIANTS:
        labs[v] = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libcamera_bwd_lab_%s.so" % v.split(":")[0]))
        labs[v].camera_bwd_lab.argtypes = _capi.load().scnerf_camera_rays_bwd.argtypes     # the same list: 21 common, 8 pointers, n, stream
        labs[v].camera_bwd_lab.restype = ctypes.c_int
    dev = torch.device("cuda:0")
    net_c, net_f, cam, query = build(dev)
    n, sc, sf = 1025, 64, 128
    lo, hi = (0, 1025) if rank == 0 else (0, 513)
    kps, idx, target, rnd = batch(n, sc, sf, dev)
    refs, events, shown = {}, {v: {"dump": 0, "sums only": 0} for v in VARIANTS}, 0
    calls = 0
    for it in range(iters):
        for m in (net_c, net_f, cam):
            for p in m.parameters(): p.grad = None
        rays_o, rays_d = get_rays_kps_use_camera(H=H, W=W, camera_model=cam, idx_in_camera_param=idx[lo:hi], kps_list=kps[lo:hi])
        rays_o.retain_grad(); rays_d.retain_grad()
        rgb, disp, acc, extras = render(H=H, W=W, chunk=1 << 15, rays=torch.stack([rays_o, rays_d]), retraw=True, camera_model=cam, mode="train",
                                        network_fn=net_c, network_fine=net_f, network_query_fn=query, N_samples=sc, N_importance=sf, perturb=1.0,
                                        raw_noise_std=1.0, use_viewdirs=True, white_bkgd=False, near=0., far=1., _randoms={k: v[lo:hi] for k, v in rnd.items()})
        loss = torch.mean((rgb - target[lo:hi]) ** 2) + torch.mean((extras["rgb0"] - target[lo:hi]) ** 2)
        loss.backward()
        c = rays_d.grad_fn.cam
        nr = int(c["n"])
        g_o, g_d = rays_o.grad.contiguous(), rays_d.grad.contiguous()
        C = int(c["extr_init"].shape[0])
        for k, v in [(k, v) for k in range(K) for v in VARIANTS]:
            lab, ref = labs[v], refs.get(v)
            d_in = torch.empty(4, device=dev); d_ex = torch.empty((C, 9), device=dev)
            d_go = torch.empty_like(c["grid_o"]); d_gd = torch.empty_like(c["grid_d"])
            ws = torch.empty(4 + 12 * C, device=dev); dump = torch.empty((nr, 32), device=dev)
            P = ops._p
            common = list(ops._cam_common(c))
            common[4] = FLAGS.get(v.split(":")[-1], 0)          # n_ext carries the lab flags
            st = lab.camera_bwd_lab(*common, P(g_o), P(g_d), P(d_in), P(d_ex), P(d_go), P(d_gd), P(ws), P(dump), nr, ops._stream())
            assert st == 0, st
            calls += 1
            cur = {"dump": dump.cpu().numpy().view(np.uint32), "intr": d_in.cpu().numpy(), "extr": d_ex.cpu().numpy(), "grid_d": d_gd.cpu().numpy(), "grid_o": d_go.cpu().numpy()}
            if ref is None:
                refs[v] = cur; continue
            moved = (cur["dump"][:, :30] != ref["dump"][:, :30])
            rel = {o: float(np.abs(cur[o] - ref[o]).max() / (np.abs(ref[o]).max() + 1e-30)) for o in ("intr", "extr", "grid_d", "grid_o")}
            if moved.any():
                events[v]["dump"] += 1
                if shown < 12:
                    shown += 1
                    rays = np.nonzero(moved.any(1))[0]
                    cols = [COLS[j] for j in np.nonzero(moved.any(0))[0]]
                    waves = sorted(set((rays // 64).tolist()))
                    first = int(rays[0]); j0 = int(np.nonzero(moved[first])[0][0])
                    a, b = ref["dump"][first, j0:j0 + 1].view(np.float32)[0], cur["dump"][first, j0:j0 + 1].view(np.float32)[0]
                    if v.startswith("v4"):
                        f32 = lambda a: a.view(np.float32)
                        print("    v4 trace, moved rays: tap00 (x, y, z) loaded now %s was %s; blend y now %s was %s; rays with tap00.y == 0 now: %d" % (
                            f32(cur["dump"][rays[:3], 0:3]).tolist(), f32(ref["dump"][rays[:3], 0:3]).tolist(), f32(cur["dump"][rays[:3], 5]).tolist(),
                            f32(ref["dump"][rays[:3], 5]).tolist(), int((f32(cur["dump"][rays, 1]) == 0).sum())), flush=True)
                    print(("rank %d iter %d call %d " + v + ": DUMP moved in %d rays (waves %s, lanes %s..), columns %s; ray %d %s %.9g -> %.9g; wave now on %s, was on %s; sums %s") % (
                        rank, it, k, len(rays), waves[:10], (rays % 64)[:8].tolist(), cols, first, COLS[j0], a, b,
                        hw_str(cur["dump"][first, 31]), hw_str(ref["dump"][first, 31]), "  ".join("%s %.1e" % kv for kv in rel.items())), flush=True)
            elif max(rel.values()) > 1e-5:
                events[v]["sums only"] += 1
                if shown < 12:
                    shown += 1
                    wv = np.nonzero(cur["dump"][::64, 30] != ref["dump"][::64, 30])[0]
                    print(("rank %d iter %d call %d " + v + ": per-ray values identical, SUMS moved: %s; waves whose reduced gk0 moved %s") % (
                        rank, it, k, "  ".join("%s %.1e" % kv for kv in rel.items()), wv.tolist()), flush=True)
    print("rank %d of %d done: %d calls (all variants); calls whose per-ray dump moved / whose sums alone moved: %s" % (rank, world, calls, events), flush=True)


if __name__ == "__main__":
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    mp.spawn(loop, args=(int(sys.argv[1]), int(sys.argv[2]), world), nprocs=world, join=True)
