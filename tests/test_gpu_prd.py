"""GPU parity of the projected-ray-distance loss (pytest -m gpu): scnerf_amd.ray_dist_loss through the
C ABI vs golden vectors of the reference's proj_ray_dist_loss_single (BASELINE config 4's extra term)."""
import types

import numpy as np
import pytest
import torch

from conftest import t
from test_emu_prd import G, NAMES, case, check_grad, truth64
from test_gpu_camera import HH, WW, M, make_camera  # noqa: F401  (M is a fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["leaf", "leaf_tight"])
def test_loss_and_input_gradients(tag):
    from scnerf_amd import ray_dist_loss as R
    d, (i0, i1) = case(tag)
    k = tag + "/"
    leaves = {n: t(d[n]).cuda().requires_grad_(True) for n in NAMES + ("K",)}
    E = t(G[k + "E"]).cuda().requires_grad_(True)
    args = types.SimpleNamespace(proj_ray_dist_threshold=d["thr"])
    loss, nm = R.proj_ray_dist_loss_single(t(d["kps0"]).cuda(), t(d["kps1"]).cuda(), i0, i1,
                                           (leaves["rays0_o"], leaves["rays0_d"]), (leaves["rays1_o"], leaves["rays1_d"]),
                                           "train", "cuda", HH, WW, args, intrinsic=leaves["K"], extrinsic=E)
    assert nm == float(G[k + "n_match"]) and isinstance(nm, float)
    assert abs(loss.item() - float(G[k + "loss"])) <= 2e-5 * float(G[k + "loss"])
    (3.0 * loss).backward()
    exact = truth64(tag)
    for n in NAMES:
        check_grad(leaves[n].grad.cpu().numpy() / 3.0, G[k + "g_" + n], exact[n], n)
    check_grad(leaves["K"].grad.cpu().numpy() / 3.0, G[k + "g_K"], exact["K"], "g_K")
    gE = E.grad.cpu().numpy() / 3.0
    check_grad(gE[[i0, i1]], G[k + "g_E"][[i0, i1]], exact["E"], "g_E")
    others = [c for c in range(gE.shape[0]) if c not in (i0, i1)]
    assert np.all(gE[others] == 0)


def test_eval_mode_and_no_valid_match():
    from scnerf_amd import ray_dist_loss as R
    d, (i0, i1) = case("leaf")
    c = {n: t(d[n]).cuda() for n in NAMES + ("K", "kps0", "kps1")}
    E = t(G["leaf/E"]).cuda()
    args = types.SimpleNamespace(proj_ray_dist_threshold=5.0)
    loss, nm = R.proj_ray_dist_loss_single(c["kps0"], c["kps1"], i0, i1, (c["rays0_o"], c["rays0_d"]),
                                           (c["rays1_o"], c["rays1_d"]), "val", "cuda", HH, WW, args,
                                           intrinsic=c["K"], extrinsic=E)
    assert nm is None
    assert abs(loss.item() - float(G["eval/loss"])) <= 2e-5 * float(G["eval/loss"])
    args.proj_ray_dist_threshold = 0.0          # nothing passes -> mean of an empty selection, nan as the reference
    loss, nm = R.proj_ray_dist_loss_single(c["kps0"], c["kps1"], i0, i1, (c["rays0_o"], c["rays0_d"]),
                                           (c["rays1_o"], c["rays1_d"]), "train", "cuda", HH, WW, args,
                                           intrinsic=c["K"], extrinsic=E)
    assert np.isnan(loss.item()) and nm == 0.0


def test_training_call_chain_through_camera_model(M):
    """run_nerf.py:536-587: key-point rays of both images from the camera model, then the loss with
    camera_model + i_map; gradients reach every camera parameter."""
    from scnerf_amd import ray_dist_loss as R
    cm, spec, _ = make_camera(M, "pinhole_rot_noise_10k_rayo_rayd", True)
    i0, i1 = (int(v) for v in G["leaf/idx"])
    i_map = G["camera/i_map"]
    k0, k1 = t(G["leaf/kps0"]).cuda(), t(G["leaf/kps1"]).cuda()
    r0 = M.gr.get_rays_kps_use_camera(HH, WW, cm, k0, idx_in_camera_param=i0)
    r1 = M.gr.get_rays_kps_use_camera(HH, WW, cm, k1, idx_in_camera_param=i1)
    np.testing.assert_allclose(r0[1].detach().cpu().numpy(), G["leaf/rays0_d"], rtol=1e-5, atol=1e-6)
    args = types.SimpleNamespace(proj_ray_dist_threshold=5.0)
    loss, nm = R.proj_ray_dist_loss_single(k0, k1, int(i_map[i0]), int(i_map[i1]), r0, r1, "train", "cuda", HH, WW,
                                           args, camera_model=cm, method="NeRF", i_map=i_map)
    assert nm == float(G["camera/n_match"])
    assert abs(loss.item() - float(G["camera/loss"])) <= 2e-5 * float(G["camera/loss"])
    # `_sync=False`: the same call without the host read -- the count comes back as a 0-dim device tensor, the loss is the same
    loss2, nm2 = R.proj_ray_dist_loss_single(k0, k1, int(i_map[i0]), int(i_map[i1]), r0, r1, "train", "cuda", HH, WW,
                                             args, camera_model=cm, method="NeRF", i_map=i_map, _sync=False)
    assert torch.is_tensor(nm2) and nm2.is_cuda and nm2.dim() == 0 and float(nm2) == nm
    assert abs(float(loss2) - float(loss)) <= 1e-6 * abs(float(loss))           # (masked means summed with atomics: last-bit order effects)
    loss.backward()
    for name in ("intrinsics_noise", "extrinsics_noise", "ray_o_noise", "ray_d_noise"):
        got = getattr(cm, name).grad.cpu().numpy()
        ref = G["camera/g_" + name]
        scale = float(np.abs(ref).max())
        assert np.isfinite(got).all()
        assert float(np.abs(got - ref).max()) <= 2e-3 * scale, name      # fp32 conditioning: see truth64()
