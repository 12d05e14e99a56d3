"""Generates tests/golden/*.npz from the UNMODIFIED reference (run in the build
container only; /root/reference does not exist on the GPU box).

    python oracle/gen_golden.py            # rewrites every fixture

Randomness inside the reference (torch.rand / torch.randn in NeRF/render.py:249,
:330, :429) is replaced by seeded tensors that are stored in the fixture, by
patching the two torch factory functions for the duration of the call; the
indices returned by torch.searchsorted (:444) are recorded the same way.  Nothing
in the reference's files is edited.

TEST INFRASTRUCTURE ONLY.
"""
import argparse
import contextlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import load_reference            # noqa: E402
from oracle import scnerf_oracle as O                   # noqa: E402
from oracle import synth                                # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


@contextlib.contextmanager
def injected_randomness(rand_queue, randn_queue, record):
    """torch.rand(shape) / torch.randn(shape) pop pre-drawn tensors; searchsorted
    results are appended to record['inds'], its inputs to record['cdf'/'u']."""
    real_rand, real_randn, real_ss = torch.rand, torch.randn, torch.searchsorted

    def fake_rand(*a, **k):
        t = rand_queue.pop(0)
        shape = tuple(a[0]) if len(a) == 1 and not isinstance(a[0], int) else tuple(a)
        assert tuple(t.shape) == shape, (t.shape, shape)
        return t.clone()

    def fake_randn(*a, **k):
        t = randn_queue.pop(0)
        shape = tuple(a[0]) if len(a) == 1 and not isinstance(a[0], int) else tuple(a)
        assert tuple(t.shape) == shape, (t.shape, shape)
        return t.clone()

    def rec_ss(cdf, u, **k):
        out = real_ss(cdf, u, **k)
        record.setdefault("cdf", []).append(cdf.detach().clone())
        record.setdefault("u", []).append(u.detach().clone())
        record.setdefault("inds", []).append(out.clone())
        return out

    torch.rand, torch.randn, torch.searchsorted = fake_rand, fake_randn, rec_ss
    try:
        yield
    finally:
        torch.rand, torch.randn, torch.searchsorted = real_rand, real_randn, real_ss


def np32(t):
    return t.detach().cpu().numpy()


def ref_network(ns, params, n_importance):
    """Reference NeRF module carrying `params` + the reference query closure
    (NeRF/create_nerf.py:42-69 without DataParallel)."""
    embed_fn, input_ch = ns.helpers.get_embedder(10, 0)
    embeddirs_fn, input_ch_views = ns.helpers.get_embedder(4, 0)
    net = ns.helpers.NeRF(D=8, W=256, input_ch=input_ch, output_ch=5 if n_importance > 0 else 4,
                          skips=[4], input_ch_views=input_ch_views, use_viewdirs=True)
    net.load_state_dict(params)
    query = lambda inputs, viewdirs, network_fn: ns.create_nerf.run_network(
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
        netchunk=1 << 20)
    return net, query


# ------------------------------------------------------------------ generators
def gen_init_check(ns):
    """Pins oracle.xavier_nerf_params(seed) == reference NeRF() under manual_seed."""
    out = {}
    for seed in (0, 3):
        torch.manual_seed(seed)
        net = ns.helpers.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4],
                              input_ch_views=27, use_viewdirs=True)
        sd = net.state_dict()
        for k in ("pts_linears.0.weight", "rgb_linear.weight", "alpha_linear.weight"):
            out["seed%d/%s" % (seed, k)] = np32(sd[k])
        out["seed%d/sumsq" % seed] = np.array(
            [float((v.double() ** 2).sum()) for v in sd.values()])
    np.savez_compressed(os.path.join(GOLDEN, "init_check.npz"), **out)


def gen_embedder(ns):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(97, 3, generator=g) * 1.3
    x[0] = 0.0
    x[1] = torch.tensor([1.5, -1.5, 0.25])
    f10, d10 = ns.helpers.get_embedder(10, 0)
    f4, d4 = ns.helpers.get_embedder(4, 0)
    np.savez_compressed(os.path.join(GOLDEN, "embedder.npz"), x=np32(x),
                        pe10=np32(f10(x)), pe4=np32(f4(x)))


def gen_mlp(ns):
    params = synth.network_params(seed=0)
    net, _ = ref_network(ns, params, 128)
    g = torch.Generator().manual_seed(5)
    emb = torch.randn(200, 90, generator=g)
    emb.requires_grad_(True)
    y = net(emb)
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    out = dict(emb=np32(emb), y=np32(y), gy=np32(gy), g_emb=np32(emb.grad))
    for k, v in net.named_parameters():
        if k.endswith(".bias") or k in ("pts_linears.0.weight", "pts_linears.5.weight",
                                        "rgb_linear.weight", "alpha_linear.weight"):
            out["g/" + k] = np32(v.grad)
        else:
            out["gnorm/" + k] = np.array(float(v.grad.double().norm()))
    np.savez_compressed(os.path.join(GOLDEN, "mlp.npz"), **out)


def gen_sample_pdf(ns):
    """KATs for the inverse-CDF sampler incl. the edge cases of SURVEY 8c(1)."""
    g = torch.Generator().manual_seed(21)
    n, m, sf = 24, 62, 128
    bins = torch.sort(torch.rand(n, m + 1, generator=g), dim=-1)[0]
    w = torch.rand(n, m, generator=g) ** 4
    w[0] = 1.0                                   # flat pdf
    w[1] = 0.0
    w[1, 17] = 1.0                               # single spike
    w[2] = 0.0                                   # all zero -> 1e-5 pad path
    w[3, :40] = 0.0                              # long empty prefix: denom < 1e-5 path
    w[4] = torch.rand(m, generator=g) * 1e-7     # everything tiny
    u = torch.rand(n, sf, generator=g)
    out = {"bins": np32(bins), "weights": np32(w)}
    for tag, det in (("rand", False), ("det", True)):
        rec = {}
        uq = [u.clone()] if not det else []
        with injected_randomness(uq, [], rec):
            s = ns.render.sample_pdf(bins, w, sf, det=det)
        out[tag + "/u"] = np32(rec["u"][0])
        out[tag + "/cdf"] = np32(rec["cdf"][0])
        out[tag + "/inds"] = rec["inds"][0].numpy()
        out[tag + "/samples"] = np32(s)
    # u exactly on cdf knots (ties must resolve like side='right')
    cdf = torch.from_numpy(out["rand/cdf"])
    u_knot = cdf[:, torch.arange(0, 63, 63 // 32)[:32]].repeat(1, 4)[:, :sf].contiguous()
    u_knot = torch.clamp(u_knot, max=float(np.nextafter(np.float32(1), np.float32(0))))
    rec = {}
    with injected_randomness([u_knot.clone()], [], rec):
        s = ns.render.sample_pdf(bins, w, sf, det=False)
    out["knot/u"] = np32(u_knot)
    out["knot/cdf"] = np32(rec["cdf"][0])
    out["knot/inds"] = rec["inds"][0].numpy()
    out["knot/samples"] = np32(s)
    np.savez_compressed(os.path.join(GOLDEN, "sample_pdf.npz"), **out)


def gen_composite(ns):
    g = torch.Generator().manual_seed(31)
    out = {}
    for tag, s in (("s64", 64), ("s192", 192)):
        n = 20
        raw = torch.randn(n, s, 4, generator=g) * 2.0
        raw[0, :, 3] = -5.0                         # zero density everywhere
        raw[1, :, 3] = 60.0                         # saturated alpha from the first sample
        raw[2, : s // 2, 3] = -1.0
        raw[2, s // 2, 3] = 1e4                      # wall in the middle
        z = torch.sort(torch.rand(n, s, generator=g), dim=-1)[0]
        z[3] = z[3, 0]                              # degenerate: all samples coincide
        rays_d = torch.randn(n, 3, generator=g)
        noise = torch.randn(n, s, generator=g)
        for wb in (False, True):
            for with_noise in (False, True):
                r = raw.clone().requires_grad_(True)
                d = rays_d.clone().requires_grad_(True)
                with injected_randomness([], [noise.clone()] if with_noise else [], {}):
                    rgb, disp, acc, w, depth = ns.render.raw2outputs(
                        r, z, d, raw_noise_std=1.0 if with_noise else 0.0, white_bkgd=wb)
                g_rgb = torch.randn(rgb.shape, generator=torch.Generator().manual_seed(7))
                g_disp = torch.randn(disp.shape, generator=torch.Generator().manual_seed(8)) * 1e-2
                g_acc = torch.randn(acc.shape, generator=torch.Generator().manual_seed(9))
                g_depth = torch.randn(acc.shape, generator=torch.Generator().manual_seed(10))
                ((rgb * g_rgb).sum() + (disp * g_disp).sum() + (acc * g_acc).sum()
                 + (depth * g_depth).sum()).backward()
                key = "%s/wb%d_n%d/" % (tag, wb, with_noise)
                out.update({key + "rgb": np32(rgb), key + "disp": np32(disp),
                            key + "acc": np32(acc), key + "weights": np32(w),
                            key + "depth": np32(depth), key + "g_raw": np32(r.grad),
                            key + "g_rays_d": np32(d.grad)})
        out.update({tag + "/raw": np32(raw), tag + "/z": np32(z), tag + "/rays_d": np32(rays_d),
                    tag + "/noise": np32(noise), tag + "/g_rgb": np32(g_rgb),
                    tag + "/g_disp": np32(g_disp), tag + "/g_acc": np32(g_acc),
                    tag + "/g_depth": np32(g_depth)})
    np.savez_compressed(os.path.join(GOLDEN, "composite.npz"), **out)


def gen_render_rays(ns):
    """Full render_rays (+ clamp of batchify_rays) at small N, outputs and grads."""
    out = {}
    cases = [
        # tag, N, S_c, S_f, perturb, raw_noise_std, lindisp, white_bkgd
        ("c64_f0_det", 24, 64, 0, 0.0, 0.0, False, False),
        ("c64_f0_pert", 24, 64, 0, 1.0, 1.0, False, False),
        ("c64_f128_pert", 24, 64, 128, 1.0, 1.0, False, False),
        ("c64_f128_det", 16, 64, 128, 0.0, 0.0, False, True),
        ("c64_f64_lindisp", 16, 64, 64, 1.0, 0.0, True, False),
    ]
    for tag, n, sc, sf, perturb, rns, lindisp, wb in cases:
        pc = synth.network_params(seed=0)
        pf = synth.network_params(seed=1)
        net_c, query = ref_network(ns, pc, sf)
        net_f, _ = ref_network(ns, pf, sf)
        rays = synth.ray_batch(n, seed=1, lindisp=lindisp)
        target = synth.target_rgb(n, seed=2)
        rnd = synth.render_randoms(n, sc, sf, seed=3)
        rays_req = rays.clone().requires_grad_(True)
        rand_q, randn_q = [], []
        if perturb > 0:
            rand_q.append(rnd["t_rand"])
        if rns > 0:
            randn_q.append(rnd["noise_c"] / 1.0)
        if sf > 0:
            if perturb > 0:
                rand_q.append(rnd["u"])
            if rns > 0:
                randn_q.append(rnd["noise_f"])
        rec = {}
        with injected_randomness(rand_q, randn_q, rec):
            ret = ns.render.batchify_rays(
                rays_req, chunk=1 << 15, network_fn=net_c, network_query_fn=query,
                N_samples=sc, retraw=True, lindisp=lindisp, perturb=perturb,
                N_importance=sf, network_fine=net_f if sf > 0 else None,
                white_bkgd=wb, raw_noise_std=rns)
        assert not rand_q and not randn_q
        loss = torch.mean((ret["rgb_map"] - target) ** 2)
        if sf > 0:
            loss = loss + torch.mean((ret["rgb0"] - target) ** 2)
        loss.backward()
        k = tag + "/"
        out[k + "cfg"] = np.array([n, sc, sf, perturb, rns, int(lindisp), int(wb)], dtype=np.float64)
        for name in ("rgb_map", "disp_map", "acc_map", "raw", "rgb0", "disp0", "acc0", "z_std"):
            if name in ret:
                out[k + name] = np32(ret[name])
        if sf > 0:
            out[k + "inds"] = rec["inds"][0].numpy()
            out[k + "cdf"] = np32(rec["cdf"][0])
        out[k + "loss"] = np.array(float(loss.detach()))
        out[k + "rays"] = np32(rays)
        out[k + "target"] = np32(target)
        for rk, rv in rnd.items():
            out[k + "rnd/" + rk] = np32(rv)
        out[k + "g_rays"] = np32(rays_req.grad)
        for netname, net in (("coarse", net_c), ("fine", net_f)):
            if netname == "fine" and sf == 0:
                continue
            for pn, v in net.named_parameters():
                if v.grad is None:
                    continue
                if pn.endswith(".bias") or pn in ("pts_linears.0.weight", "rgb_linear.weight",
                                                  "alpha_linear.weight"):
                    out[k + "g/%s/%s" % (netname, pn)] = np32(v.grad)
                else:
                    out[k + "gnorm/%s/%s" % (netname, pn)] = np.array(float(v.grad.double().norm()))
    np.savez_compressed(os.path.join(GOLDEN, "render_rays.npz"), **out)


def gen_camera(ns):
    out = {}
    H, W = 378, 504
    for tag, cls_name, mult in (("plain_add", "PinholeModelRotNoiseLearning10kRayoRayd", False),
                                ("plain_mul", "PinholeModelRotNoiseLearning10kRayoRayd", True),
                                ("dist_mul", "PinholeModelRotNoiseLearning10kRayoRaydDistortion", True)):
        spec = synth.camera_spec(H, W, n_cams=5, seed=4, multiplicative=mult)
        args = types.SimpleNamespace(
            camera_model="pinhole_rot_noise_10k_rayo_rayd", grid_size=10,
            ray_o_noise_scale=spec["ray_o_noise_scale"], ray_d_noise_scale=spec["ray_d_noise_scale"],
            extrinsics_noise_scale=spec["extrinsics_noise_scale"],
            intrinsics_noise_scale=spec["intrinsics_noise_scale"], multiplicative_noise=mult,
            distortion_noise_scale=1e-2)
        cls = getattr(ns.camera_model, cls_name)
        cm = cls(spec["K_init"], list(spec["poses"].numpy()), args, H, W)
        with torch.no_grad():
            cm.intrinsics_noise.copy_(spec["intrinsics_noise"])
            cm.extrinsics_noise.copy_(spec["extrinsics_noise"])
            cm.ray_o_noise.copy_(spec["ray_o_noise"])
            if cm.ray_d_noise.data_ptr() != cm.ray_o_noise.data_ptr():
                cm.ray_d_noise.copy_(spec["ray_d_noise"])
        kps, idx = synth.keypoints(H, W, 64, n_cams=5, seed=6)
        ro, rd = ns.get_rays.get_rays_kps_use_camera(H, W, cm, kps, idx_in_camera_param=idx)
        g_o = torch.randn(ro.shape, generator=torch.Generator().manual_seed(12))
        g_d = torch.randn(rd.shape, generator=torch.Generator().manual_seed(13))
        ((ro * g_o).sum() + (rd * g_d).sum()).backward()
        k = tag + "/"
        out.update({k + "rays_o": np32(ro), k + "rays_d": np32(rd), k + "g_o": np32(g_o),
                    k + "g_d": np32(g_d), k + "kps": np32(kps), k + "idx": idx.numpy(),
                    k + "K": np32(cm.get_intrinsic()), k + "E": np32(cm.get_extrinsic()),
                    k + "g_intrinsics_noise": np32(cm.intrinsics_noise.grad),
                    k + "g_extrinsics_noise": np32(cm.extrinsics_noise.grad),
                    k + "g_ray_o_noise": np32(cm.ray_o_noise.grad),
                    k + "g_ray_d_noise": np32(cm.ray_d_noise.grad),
                    k + "aliased": np.array(int(cm.ray_d_noise.data_ptr() == cm.ray_o_noise.data_ptr()))})
        # shared-extrinsic branch (extrinsic.dim()==2) + NDC through the camera model
        for p in cm.parameters():
            p.grad = None
        E = cm.get_extrinsic()[2]
        ro2, rd2 = ns.get_rays.get_rays_kps_use_camera(H, W, cm, kps, extrinsic=E)
        no, nd = ns.render.ndc_rays_camera(H, W, cm, 1.0, ro2, rd2)
        ((no * g_o).sum() + (nd * g_d).sum()).backward()
        out.update({k + "shared/rays_o": np32(ro2), k + "shared/rays_d": np32(rd2),
                    k + "shared/ndc_o": np32(no), k + "shared/ndc_d": np32(nd),
                    k + "shared/g_intrinsics_noise": np32(cm.intrinsics_noise.grad),
                    k + "shared/g_extrinsics_noise": np32(cm.extrinsics_noise.grad),
                    k + "shared/g_ray_o_noise": np32(cm.ray_o_noise.grad)})
    # pinhole without camera model + plain ndc_rays
    c2w = synth.camera_spec(H, W, n_cams=5, seed=4)["poses"][1]
    kps, _ = synth.keypoints(H, W, 64, n_cams=5, seed=6)
    kps3 = torch.cat([kps, torch.ones_like(kps[:, :1])], -1)
    ro, rd = ns.get_rays.get_rays_kps_no_camera(H, W, 400.0, c2w, kps3)
    no, nd = ns.render.ndc_rays(H, W, 400.0, 1.0, ro, rd)
    out.update({"pinhole/c2w": np32(c2w), "pinhole/kps": np32(kps), "pinhole/rays_o": np32(ro),
                "pinhole/rays_d": np32(rd), "pinhole/ndc_o": np32(no), "pinhole/ndc_d": np32(nd)})
    np.savez_compressed(os.path.join(GOLDEN, "camera.npz"), **out)


def _ref_camera(ns, H, W, n_cams=5, mult=True):
    spec = synth.camera_spec(H, W, n_cams=n_cams, seed=4, multiplicative=mult)
    args = types.SimpleNamespace(
        camera_model="pinhole_rot_noise_10k_rayo_rayd", grid_size=10,
        ray_o_noise_scale=spec["ray_o_noise_scale"], ray_d_noise_scale=spec["ray_d_noise_scale"],
        extrinsics_noise_scale=spec["extrinsics_noise_scale"],
        intrinsics_noise_scale=spec["intrinsics_noise_scale"], multiplicative_noise=mult,
        distortion_noise_scale=1e-2)
    cm = ns.camera_model.PinholeModelRotNoiseLearning10kRayoRayd(
        spec["K_init"], list(spec["poses"].numpy()), args, H, W)
    with torch.no_grad():
        cm.intrinsics_noise.copy_(spec["intrinsics_noise"])
        cm.extrinsics_noise.copy_(spec["extrinsics_noise"])
        cm.ray_o_noise.copy_(spec["ray_o_noise"])
        cm.ray_d_noise.copy_(spec["ray_d_noise"])
    return cm


def gen_prd(ns):
    """proj_ray_dist_loss_single (model/ray_dist_loss.py:22) on synthetic matches of two cameras of the
    synthetic rig: (leaf) every float input a leaf tensor -> loss, n_match and all input gradients;
    (camera) the run_nerf.py:536-587 call chain through the camera model -> parameter gradients;
    (eval) mode "val"."""
    out = {}
    H, W = 378, 504
    cm = _ref_camera(ns, H, W)
    prd = ns.ray_dist_loss.proj_ray_dist_loss_single
    i0, i1 = 1, 3
    with torch.no_grad():
        K = cm.get_intrinsic().clone()
        E = cm.get_extrinsic().clone()
    k0, k1 = synth.matched_keypoints(H, W, K, E[i0], E[i1], 300, seed=8)
    args = types.SimpleNamespace(proj_ray_dist_threshold=5.0)
    for tag, thr in (("leaf", 5.0), ("leaf_tight", 1.0)):
        args.proj_ray_dist_threshold = thr
        with torch.no_grad():
            r0 = ns.get_rays.get_rays_kps_use_camera(H, W, cm, k0, idx_in_camera_param=i0)
            r1 = ns.get_rays.get_rays_kps_use_camera(H, W, cm, k1, idx_in_camera_param=i1)
        leaves = [t.clone().requires_grad_(True) for t in (r0[0], r0[1], r1[0], r1[1], K, E)]
        loss, nm = prd(k0, k1, i0, i1, (leaves[0], leaves[1]), (leaves[2], leaves[3]), "train", "cpu", H, W, args,
                       intrinsic=leaves[4], extrinsic=leaves[5], method="NeRF")
        loss.backward()
        k = tag + "/"
        out.update({k + "kps0": np32(k0), k + "kps1": np32(k1), k + "idx": np.array([i0, i1]),
                    k + "threshold": np.array(thr, np.float32), k + "loss": np32(loss), k + "n_match": np.array(nm),
                    k + "rays0_o": np32(leaves[0]), k + "rays0_d": np32(leaves[1]), k + "rays1_o": np32(leaves[2]),
                    k + "rays1_d": np32(leaves[3]), k + "K": np32(leaves[4]), k + "E": np32(leaves[5]),
                    k + "g_rays0_o": np32(leaves[0].grad), k + "g_rays0_d": np32(leaves[1].grad),
                    k + "g_rays1_o": np32(leaves[2].grad), k + "g_rays1_d": np32(leaves[3].grad),
                    k + "g_K": np32(leaves[4].grad), k + "g_E": np32(leaves[5].grad)})
    # through the camera model, as the training loop calls it (i_map = i_train maps image id -> camera slot)
    args.proj_ray_dist_threshold = 5.0
    i_map = np.array([10, 11, 12, 13, 14])
    r0 = ns.get_rays.get_rays_kps_use_camera(H, W, cm, k0, idx_in_camera_param=i0)
    r1 = ns.get_rays.get_rays_kps_use_camera(H, W, cm, k1, idx_in_camera_param=i1)
    loss, nm = prd(k0, k1, int(i_map[i0]), int(i_map[i1]), r0, r1, "train", "cpu", H, W, args,
                   camera_model=cm, method="NeRF", i_map=i_map)
    loss.backward()
    out.update({"camera/loss": np32(loss), "camera/n_match": np.array(nm), "camera/i_map": i_map,
                "camera/g_intrinsics_noise": np32(cm.intrinsics_noise.grad),
                "camera/g_extrinsics_noise": np32(cm.extrinsics_noise.grad),
                "camera/g_ray_o_noise": np32(cm.ray_o_noise.grad),
                "camera/g_ray_d_noise": np32(cm.ray_d_noise.grad)})
    # eval mode (no gradient): the reference indexes extrinsic[[[i0, i1]]] (:86,94)
    try:
        with torch.no_grad():
            r0 = ns.get_rays.get_rays_kps_use_camera(H, W, cm, k0, idx_in_camera_param=i0)
            r1 = ns.get_rays.get_rays_kps_use_camera(H, W, cm, k1, idx_in_camera_param=i1)
            loss_e, _ = prd(k0, k1, i0, i1, r0, r1, "val", "cpu", H, W, args, intrinsic=K, extrinsic=E, method="NeRF")
        out["eval/loss"] = np32(loss_e)
    except Exception as exc:                     # recorded so the tests know the pin is absent
        print("  eval mode of the reference failed here:", type(exc).__name__, exc)
        out["eval/unavailable"] = np.array(1)
    np.savez_compressed(os.path.join(GOLDEN, "prd.npz"), **out)


def gen_rowsum(ns):
    """Pins the explicit ATen-AVX512 row-sum restatement against torch.sum here."""
    g = torch.Generator().manual_seed(41)
    w = torch.rand(512, 62, generator=g) ** 3 + 1e-5
    np.savez_compressed(os.path.join(GOLDEN, "rowsum.npz"), w=np32(w),
                        tot=np32(torch.sum(w, -1)),
                        capability=np.array(torch.backends.cpu.get_cpu_capability()))


def gen_optimizer(ns):
    """K steps of the reference's CustomAdamOptimizer (weight decay on the trailing ray-noise tensors
    only, learning rate following run_nerf.py's schedule) and of torch.optim.Adam."""
    out = {}
    g = torch.Generator().manual_seed(51)
    shapes = [(37, 5), (64,), (3, 4, 3), (3, 4, 3)]          # two "network" tensors + ray_o / ray_d grids
    p0 = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) * (0.1 + k) for s in shapes] for k in range(4)]
    args = types.SimpleNamespace(camera_model="pinhole_rot_noise_10k_rayo_rayd")
    for tag, wd in (("custom_wd", 0.1), ("custom_nowd", 0.0), ("adam", None)):
        ps = [torch.nn.Parameter(x.clone()) for x in p0]
        if wd is None:
            opt = torch.optim.Adam(ps, lr=5e-4, betas=(0.9, 0.999))
        else:
            opt = ns.create_nerf.CustomAdamOptimizer(params=ps, lr=5e-4, betas=(0.9, 0.999), weight_decay=wd,
                                                     H=12, W=16, args=args)
        for k in range(4):
            for pp, gg in zip(ps, grads[k]):
                pp.grad = gg.clone()
            opt.step()
            new_lr = 5e-4 * (0.1 ** ((k + 1) / (250 * 1000)))
            for grp in opt.param_groups:
                grp["lr"] = new_lr
            for i, pp in enumerate(ps):
                out["%s/step%d/p%d" % (tag, k, i)] = np32(pp).copy()      # (np32 aliases the live parameter)
    for i, x in enumerate(p0):
        out["p0/%d" % i] = np32(x)
    for k in range(4):
        for i, x in enumerate(grads[k]):
            out["grad%d/%d" % (k, i)] = np32(x)
    np.savez_compressed(os.path.join(GOLDEN, "optimizer.npz"), **out)


def gen_checkpoint(ns):
    """A checkpoint written exactly as run_nerf.py:626-641 writes it (DataParallel-wrapped networks ->
    'module.' keys, the optimizer's state_dict, the camera model's state_dict) after two optimizer
    steps of the reference, + the parameters after a third step with recorded gradients: what a
    restored run must reproduce.  Tiny networks (D=3, W=16) keep the fixture small."""
    H, W = 20, 30
    spec = synth.camera_spec(H, W, n_cams=4, seed=14, multiplicative=True)
    args = types.SimpleNamespace(
        camera_model="pinhole_rot_noise_10k_rayo_rayd", grid_size=10,
        ray_o_noise_scale=spec["ray_o_noise_scale"], ray_d_noise_scale=spec["ray_d_noise_scale"],
        extrinsics_noise_scale=spec["extrinsics_noise_scale"],
        intrinsics_noise_scale=spec["intrinsics_noise_scale"], multiplicative_noise=True)
    torch.manual_seed(5)

    def net():
        return torch.nn.DataParallel(ns.helpers.NeRF(D=3, W=16, input_ch=63, output_ch=5, skips=[4],
                                                     input_ch_views=27, use_viewdirs=True))
    model, model_fine = net(), net()
    cm = ns.camera_model.PinholeModelRotNoiseLearning10kRayoRayd(spec["K_init"], list(spec["poses"].numpy()), args, H, W)
    grad_vars = list(model.parameters()) + list(model_fine.parameters()) + list(cm.parameters())
    opt = ns.create_nerf.CustomAdamOptimizer(params=grad_vars, lr=5e-4, betas=(0.9, 0.999), weight_decay=0.1,
                                             H=H, W=W, args=args)
    g = torch.Generator().manual_seed(77)

    def step(k):
        gs = []
        for p in grad_vars:
            if p.requires_grad:
                p.grad = torch.randn(p.shape, generator=g) * (0.05 * (k + 1))
                gs.append(p.grad.clone())
        opt.step()
        for grp in opt.param_groups:
            grp["lr"] = 5e-4 * (0.1 ** ((k + 1) / (250 * 1000)))
        return gs
    step(0)
    step(1)
    save_dict = {"global_step": 2, "network_fn_state_dict": model.state_dict(),
                 "network_fine_state_dict": model_fine.state_dict(), "optimizer_state_dict": opt.state_dict(),
                 "camera_model": cm.state_dict()}
    torch.save(save_dict, os.path.join(GOLDEN, "ref_ckpt.tar"))
    gs = step(2)
    out = {"lr_after_reload": np.array(opt.param_groups[0]["lr"])}
    trainable = [p for p in grad_vars if p.requires_grad]
    for i, (p, gg) in enumerate(zip(trainable, gs)):
        out["grad/%d" % i] = np32(gg)
        out["after/%d" % i] = np32(p).copy()
    out["n_trainable"] = np.array(len(trainable))
    np.savez_compressed(os.path.join(GOLDEN, "ref_ckpt_after.npz"), **out)


@contextlib.contextmanager
def injected_uniforms(queue):
    """torch.rand(*shape) and torch.rand_like(x) pop pre-drawn tensors (NeRF++'s perturb_samples /
    sample_pdf draw this way, ddp_train_nerf.py:77,108)."""
    real_rand, real_like = torch.rand, torch.rand_like

    def fake_rand(*a, **k):
        t = queue.pop(0)
        shape = tuple(a[0]) if len(a) == 1 and not isinstance(a[0], int) else tuple(a)
        assert tuple(t.shape) == shape, (t.shape, shape)
        return t.clone()

    def fake_like(x, **k):
        t = queue.pop(0)
        assert t.shape == x.shape, (t.shape, x.shape)
        return t.clone()
    torch.rand, torch.rand_like = fake_rand, fake_like
    try:
        yield
    finally:
        torch.rand, torch.rand_like = real_rand, real_like


def _projections(named_grads, seed=99):
    """Per tensor: L2 norm and the dot product with a seeded Gaussian vector -- a gradient fingerprint
    that pins every entry's contribution without storing the (multi-MB) gradient itself."""
    from oracle.nerfpp_oracle import grad_fingerprint
    return {name: np.array(grad_fingerprint(name, g, seed)) for name, g in named_grads}


def gen_nerfpp(ns_unused):
    """NeRF++, from the reference's own nerfplusplus/ code: init pin,
    KATs of intersect_sphere / perturb_samples / sample_pdf / depth2pts_outside, NerfNet.forward with
    gradients, a two-level cascade training step with injected uniforms, and the pixel-centre ray
    generator with and without radial distortion."""
    from oracle.ref_import import load_nerfpp
    npp = load_nerfpp()
    out = {}
    args = types.SimpleNamespace(max_freq_log2=10, max_freq_log2_viewdirs=4, netdepth=8, netwidth=256,
                                 use_viewdirs=True)

    def make_net(seed):
        torch.manual_seed(seed)
        return npp.ddp_model.NerfNet(args)
    # --- init pin: synthetic.nerfpp_params(seed) == reference construction
    net = make_net(777)
    mine = synth.nerfpp_params(777)
    for k, v in net.state_dict().items():
        assert torch.equal(v, mine[k]), k
    out["init/abs_sums"] = np.array([float(v.double().abs().sum()) for v in net.state_dict().values()])
    out["init/first8_fg0"] = np32(net.state_dict()["fg_net.base_layers.0.0.weight"].reshape(-1)[:8])
    out["init/first8_bg5"] = np32(net.state_dict()["bg_net.base_layers.5.0.weight"].reshape(-1)[:8])

    # --- function KATs
    n = 48
    o, d, near = synth.nerfpp_rays(n, seed=21)
    out["kat/ray_o"], out["kat/ray_d"] = np32(o), np32(d)
    far = npp.train.intersect_sphere(o, d)
    out["kat/far"] = np32(far)
    g = torch.Generator().manual_seed(31)
    z = torch.sort(torch.rand(n, 24, generator=g), -1)[0]
    t_rand = torch.rand(n, 24, generator=g)
    with injected_uniforms([t_rand]):
        out["kat/perturbed"] = np32(npp.train.perturb_samples(z))
    out["kat/z"], out["kat/t_rand"] = np32(z), np32(t_rand)
    bins = torch.sort(torch.rand(n, 24, generator=g) * 3, -1)[0]
    w = torch.rand(n, 23, generator=g) ** 3
    w[0] = 0.0                                   # all-zero row: only the TINY floor remains
    w[1] = 0.0; w[1, 7] = 5.0                    # one peaked bin
    w[2, :5] = 0.0                               # leading zeros
    w[3, -6:] = 0.0                              # trailing zeros
    u = torch.rand(n, 40, generator=g)
    u[:, 0] = 0.0
    u[:, 1] = 1.0 - 1e-7
    u[4] = torch.linspace(0, 1, 40)              # the deterministic draw incl. u == 1 exactly
    with injected_uniforms([u]):
        out["kat/pdf_samples"] = np32(npp.train.sample_pdf(bins, w, 40, det=False))
    out["kat/pdf_det"] = np32(npp.train.sample_pdf(bins, w, 40, det=True))
    out["kat/bins"], out["kat/weights"], out["kat/u"] = np32(bins), np32(w), np32(u)
    depth = torch.rand(n, 16, generator=g) * 0.98 + 0.01
    depth[:, 0] = 1.0
    depth[:, 1] = 1e-3
    oo, dd = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    pts, dreal = npp.ddp_model.depth2pts_outside(oo[:, None].expand(n, 16, 3), dd[:, None].expand(n, 16, 3), depth)
    gp = torch.randn(pts.shape, generator=g)
    (pts * gp).sum().backward()
    out.update({"kat/bg_depth": np32(depth), "kat/bg_pts": np32(pts), "kat/bg_depth_real": np32(dreal),
                "kat/bg_g_pts": np32(gp), "kat/bg_g_o": np32(oo.grad), "kat/bg_g_d": np32(dd.grad)})

    # --- one NerfNet.forward with gradients (uneven sample counts, 40 rays)
    n = 40
    o, d, near = synth.nerfpp_rays(n, seed=23)
    net = make_net(778)
    oo, dd = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    far = npp.train.intersect_sphere(oo, dd)
    gz = torch.Generator().manual_seed(41)
    frac = torch.sort(torch.rand(n, 48, generator=gz), -1)[0]
    fg_z = near[:, None] + frac * (far - near)[:, None]
    bg_z = torch.sort(torch.rand(n, 40, generator=gz), -1)[0]
    ret = net(oo, dd, far, fg_z, bg_z)
    target = torch.rand(n, 3, generator=gz)
    gw = torch.randn(n, 48, generator=gz) * 1e-2
    loss = ((ret["rgb"] - target) ** 2).mean() + (ret["fg_weights"] * gw).sum() + ret["bg_depth"].mean() * 0.1 \
        + ret["fg_depth"].mean() * 0.1
    loss.backward()
    k = "fwd/"
    out.update({k + "ray_o": np32(o), k + "ray_d": np32(d), k + "frac": np32(frac), k + "bg_z": np32(bg_z),
                k + "target": np32(target), k + "gw": np32(gw), k + "loss": np32(loss), k + "far": np32(far),
                k + "g_ray_o": np32(oo.grad), k + "g_ray_d": np32(dd.grad)})
    for name, v in ret.items():
        out[k + "ret/" + name] = np32(v)
    for name, v in _projections([(a, b.grad) for a, b in net.named_parameters()]).items():
        out[k + "gproj/" + name] = v
    for name in ("fg_net.base_layers.0.0.weight", "bg_net.base_layers.0.0.weight", "bg_net.sigma_layers.0.weight",
                 "fg_net.rgb_layers.2.weight", "fg_net.rgb_layers.2.bias", "bg_net.base_remap_layers.0.bias"):
        out[k + "g/" + name] = np32(dict(net.named_parameters())[name].grad)

    # --- two-level cascade training step (ddp_train_nerf.py:430-489), 64 then 128 extra samples
    n, s0, s1 = 32, 64, 128
    o, d, near = synth.nerfpp_rays(n, seed=25)
    rnd = synth.nerfpp_randoms(n, s0, s1, seed=26)
    nets = [make_net(779), make_net(780)]
    target = torch.rand(n, 3, generator=torch.Generator().manual_seed(27))
    oo, dd = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    with injected_uniforms([rnd["t_fg"], rnd["t_bg"], rnd["u_fg"], rnd["u_bg"]]):
        far = npp.train.intersect_sphere(oo, dd)
        step = (far - near) / (s0 - 1)
        fg_depth = torch.stack([near + i * step for i in range(s0)], dim=-1)
        fg_depth = npp.train.perturb_samples(fg_depth)
        bg_depth = torch.linspace(0., 1., s0).view(1, s0).expand(n, s0)
        bg_depth = npp.train.perturb_samples(bg_depth)
        ret0 = nets[0](oo, dd, far, fg_depth, bg_depth)
        loss = ((ret0["rgb"] - target) ** 2).mean()
        fg_w = ret0["fg_weights"].clone().detach()
        fg_mid = .5 * (fg_depth[..., 1:] + fg_depth[..., :-1])
        fg_s = npp.train.sample_pdf(bins=fg_mid, weights=fg_w[..., 1:-1], N_samples=s1, det=False)
        fg_depth1, _ = torch.sort(torch.cat((fg_depth, fg_s), dim=-1))
        bg_w = ret0["bg_weights"].clone().detach()
        bg_mid = .5 * (bg_depth[..., 1:] + bg_depth[..., :-1])
        bg_s = npp.train.sample_pdf(bins=bg_mid, weights=bg_w[..., 1:-1], N_samples=s1, det=False)
        bg_depth1, _ = torch.sort(torch.cat((bg_depth, bg_s), dim=-1))
        ret1 = nets[1](oo, dd, far, fg_depth1, bg_depth1)
        loss = loss + ((ret1["rgb"] - target) ** 2).mean()
    loss.backward()
    k = "step/"
    out.update({k + "ray_o": np32(o), k + "ray_d": np32(d), k + "target": np32(target), k + "loss": np32(loss),
                k + "rgb0": np32(ret0["rgb"]), k + "rgb1": np32(ret1["rgb"]), k + "fg_depth0": np32(fg_depth),
                k + "bg_depth0": np32(bg_depth), k + "fg_depth1": np32(fg_depth1), k + "bg_depth1": np32(bg_depth1),
                k + "bg_lambda1": np32(ret1["bg_lambda"]), k + "fg_depth_map1": np32(ret1["fg_depth"]),
                k + "g_ray_o": np32(oo.grad), k + "g_ray_d": np32(dd.grad)})
    for lvl, nn_ in enumerate(nets):
        for name, v in _projections([(a, b.grad) for a, b in nn_.named_parameters()]).items():
            out[k + "gproj%d/" % lvl + name] = v

    # --- ray generator (A18): both camera model classes
    H, W = 60, 80
    base = load_reference()
    for tag, cls_name in (("plain", "PinholeModelRotNoiseLearning10kRayoRayd"),
                          ("dist", "PinholeModelRotNoiseLearning10kRayoRaydDistortion")):
        spec = synth.camera_spec(H, W, n_cams=4, seed=33, multiplicative=True, focal=70.0)
        cargs = types.SimpleNamespace(
            camera_model="x", grid_size=10, ray_o_noise_scale=spec["ray_o_noise_scale"],
            ray_d_noise_scale=spec["ray_d_noise_scale"], extrinsics_noise_scale=spec["extrinsics_noise_scale"],
            intrinsics_noise_scale=spec["intrinsics_noise_scale"], multiplicative_noise=True,
            distortion_noise_scale=1e-1)
        cls = getattr(base.camera_model, cls_name)
        extra = (np.array([0.05, -0.02], np.float32),) if tag == "dist" else ()
        cm = cls(spec["K_init"], list(spec["poses"].numpy()), cargs, H, W, *extra)
        with torch.no_grad():
            cm.intrinsics_noise.copy_(spec["intrinsics_noise"])
            cm.extrinsics_noise.copy_(spec["extrinsics_noise"])
            cm.ray_o_noise.copy_(spec["ray_o_noise"])
            if cm.ray_d_noise.data_ptr() != cm.ray_o_noise.data_ptr():
                cm.ray_d_noise.copy_(spec["ray_d_noise"])
            if tag == "dist":
                cm.distortion_noise.copy_(torch.tensor([0.3, -0.2]))
        sel = torch.randint(0, H * W, (64,), generator=torch.Generator().manual_seed(34))
        sel[0], sel[1] = 0, H * W - 1
        ro, rd, dep = npp.rays.render_ray_from_camera(cm, 2, sel, "cpu")
        g_o = torch.randn(ro.shape, generator=torch.Generator().manual_seed(35))
        g_d = torch.randn(rd.shape, generator=torch.Generator().manual_seed(36))
        ((ro * g_o).sum() + (rd * g_d).sum()).backward()
        k = "rays_%s/" % tag
        out.update({k + "select": sel.numpy(), k + "rays_o": np32(ro), k + "rays_d": np32(rd), k + "depth": np32(dep),
                    k + "g_o": np32(g_o), k + "g_d": np32(g_d)})
        for name, prm in cm.named_parameters():
            if prm.grad is not None:
                out[k + "g_" + name] = np32(prm.grad)
        if tag == "dist":
            out[k + "k"] = extra[0]
    np.savez_compressed(os.path.join(GOLDEN, "nerfpp.npz"), **out)


def gen_nerfpp_sampler(ns_unused):
    """The NeRF++ sampler at the main path's standard (nerfplusplus/ddp_train_nerf.py:83-132): the cumulated pdf and the
    comparison-count indices (:95-98, :113) beside the samples.  The reference's sample_pdf returns the samples only;
    cdf / indices are the oracle's restatement of the same torch expressions, ACCEPTED ONLY IF the samples it forms from
    them equal the reference function's output bit for bit (asserted here, on every vector).  Also the level-0 weights of
    the cascade step (gen_nerfpp's `step/` case, re-run: its refined depths must reproduce the committed golden bit for
    bit) so that the GPU sampler can be fed the reference's own weights, and level 1's full output on the reference's
    refined depths."""
    from oracle import nerfpp_oracle as NO
    from oracle.ref_import import load_nerfpp
    npp = load_nerfpp()
    old = dict(np.load(os.path.join(GOLDEN, "nerfpp.npz")))
    out = {}
    bins, w, u = (torch.from_numpy(old[k]) for k in ("kat/bins", "kat/weights", "kat/u"))
    s, cdf, below, above = NO.sample_pdf_state(bins, w, u)
    with injected_uniforms([u]):
        ref = npp.train.sample_pdf(bins, w, 40, det=False)
    assert torch.equal(s, ref) and np.array_equal(np32(ref), old["kat/pdf_samples"])
    out["kat/cdf"], out["kat/above"], out["kat/below"] = np32(cdf), above.numpy().astype(np.int32), below.numpy().astype(np.int32)
    u_det = torch.linspace(0., 1., 40).expand(bins.shape[0], 40)
    s, cdf_d, below, above = NO.sample_pdf_state(bins, w, u_det)
    assert torch.equal(s, npp.train.sample_pdf(bins, w, 40, det=True)) and torch.equal(cdf_d, cdf)
    out["kat/above_det"] = above.numpy().astype(np.int32)

    args = types.SimpleNamespace(max_freq_log2=10, max_freq_log2_viewdirs=4, netdepth=8, netwidth=256, use_viewdirs=True)

    def make_net(seed):
        torch.manual_seed(seed)
        return npp.ddp_model.NerfNet(args)
    n, s0, s1 = 32, 64, 128
    o, d, near = synth.nerfpp_rays(n, seed=25)
    rnd = synth.nerfpp_randoms(n, s0, s1, seed=26)
    nets = [make_net(779), make_net(780)]
    target = torch.rand(n, 3, generator=torch.Generator().manual_seed(27))
    oo, dd = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    with injected_uniforms([rnd["t_fg"], rnd["t_bg"], rnd["u_fg"], rnd["u_bg"]]):
        far = npp.train.intersect_sphere(oo, dd)
        step = (far - near) / (s0 - 1)
        fg_depth = torch.stack([near + i * step for i in range(s0)], dim=-1)
        fg_depth = npp.train.perturb_samples(fg_depth)
        bg_depth = torch.linspace(0., 1., s0).view(1, s0).expand(n, s0)
        bg_depth = npp.train.perturb_samples(bg_depth)
        ret0 = nets[0](oo, dd, far, fg_depth, bg_depth)
        fg_w = ret0["fg_weights"].clone().detach()
        fg_mid = .5 * (fg_depth[..., 1:] + fg_depth[..., :-1])
        fg_s = npp.train.sample_pdf(bins=fg_mid, weights=fg_w[..., 1:-1], N_samples=s1, det=False)
        fg_depth1, _ = torch.sort(torch.cat((fg_depth, fg_s), dim=-1))
        bg_w = ret0["bg_weights"].clone().detach()
        bg_mid = .5 * (bg_depth[..., 1:] + bg_depth[..., :-1])
        bg_s = npp.train.sample_pdf(bins=bg_mid, weights=bg_w[..., 1:-1], N_samples=s1, det=False)
        bg_depth1, _ = torch.sort(torch.cat((bg_depth, bg_s), dim=-1))
        ret1 = nets[1](oo, dd, far, fg_depth1, bg_depth1)
    assert np.array_equal(np32(fg_depth1), old["step/fg_depth1"]) and np.array_equal(np32(bg_depth1), old["step/bg_depth1"])
    assert np.array_equal(np32(ret1["rgb"]), old["step/rgb1"])
    k = "step/"
    for tag, mid, wts, uu, smp in (("fg", fg_mid, fg_w, rnd["u_fg"], fg_s), ("bg", bg_mid, bg_w, rnd["u_bg"], bg_s)):
        s, cdf, below, above = NO.sample_pdf_state(mid.detach(), wts[..., 1:-1], uu)
        assert torch.equal(s, smp.detach()), tag
        out.update({k + tag + "_w0": np32(wts), k + tag + "_mid": np32(mid), k + tag + "_cdf": np32(cdf),
                    k + tag + "_above": above.numpy().astype(np.int32), k + tag + "_samples": np32(smp)})
    for name, v in ret0.items():
        out[k + "ret0/" + name] = np32(v)
    for name, v in ret1.items():
        out[k + "ret1/" + name] = np32(v)
    # level 1 alone: loss on ret1 only, gradients of its network (fingerprints) -- what the GPU must reproduce when it is
    # handed the reference's refined depths
    loss1 = ((ret1["rgb"] - target) ** 2).mean()
    loss1.backward()
    out[k + "loss1"] = np32(loss1)
    for name, v in _projections([(a, b.grad) for a, b in nets[1].named_parameters()]).items():
        out[k + "gproj1_alone/" + name] = v
    np.savez_compressed(os.path.join(GOLDEN, "nerfpp_sampler.npz"), **out)


def gen_prd_filter(ns):
    """filter_matches_with_gt (model/prd_evaluation.py:189-332) -- the reference's own function, executed from
    where it lies (its module imports cv2 / SuperGlue at the top, so only the function is taken) -- on synthetic
    matches: pixel noise, 25 % unrelated pairs, sub-threshold and beyond-threshold errors, points behind a camera."""
    from oracle.ref_import import REF_ROOT, _functions_from_source
    fn_ns = {"torch": torch, "np": np}
    _functions_from_source(os.path.join(REF_ROOT, "model", "prd_evaluation.py"), ["filter_matches_with_gt"], fn_ns)
    filt = fn_ns["filter_matches_with_gt"]
    H, W = 120, 160
    spec = synth.camera_spec(H, W, n_cams=3, seed=11, focal=140.0)
    K, E = spec["K_init"], spec["poses"]
    out = {"H": np.array(H), "W": np.array(W), "K": np32(K), "E": np32(E)}
    for tag, noise, seed in (("tight", 0.3, 8), ("loose", 1.2, 9)):
        k0, k1 = synth.matched_keypoints(H, W, K, E[0], E[1], 300, seed=seed, noise_px=noise)
        k0, k1 = k0.round(), k1.round()                      # detected key points are pixel centres; rays use .long()
        r0 = ns.get_rays.get_rays_kps_no_camera(H=H, W=W, focal=K[0][0], extrinsic=E[0], kps_list=k0)
        r1 = ns.get_rays.get_rays_kps_no_camera(H=H, W=W, focal=K[0][0], extrinsic=E[1], kps_list=k1)
        keep = filt(kps0_list=k0, kps1_list=k1, H=H, W=W, gt_intrinsic=K, gt_extrinsic=E[[0, 1]], rays0=r0, rays1=r1,
                    args=None, device="cpu", method="NeRF")
        k = tag + "/"
        out.update({k + "kps0": np32(k0), k + "kps1": np32(k1), k + "rays0_o": np32(r0[0].contiguous()),
                    k + "rays0_d": np32(r0[1]), k + "rays1_o": np32(r1[0].contiguous()), k + "rays1_d": np32(r1[1]),
                    k + "keep": keep.numpy()})
        print(tag, "kept", int(keep.sum()), "of", keep.numel())
    np.savez_compressed(os.path.join(GOLDEN, "prd_filter.npz"), **out)


ALL = dict(optimizer=gen_optimizer, prd_filter=gen_prd_filter, init=gen_init_check, embedder=gen_embedder, mlp=gen_mlp, sample_pdf=gen_sample_pdf,
           composite=gen_composite, render_rays=gen_render_rays, camera=gen_camera,
           rowsum=gen_rowsum, prd=gen_prd, checkpoint=gen_checkpoint, nerfpp=gen_nerfpp, nerfpp_sampler=gen_nerfpp_sampler)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=list(ALL))
    a = ap.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(8)
    ns = load_reference()
    for name in a.which:
        print("generating", name)
        ALL[name](ns)
    for f in sorted(os.listdir(GOLDEN)):
        print("%9d  %s" % (os.path.getsize(os.path.join(GOLDEN, f)), f))
