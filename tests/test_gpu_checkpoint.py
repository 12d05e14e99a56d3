"""Checkpoint interop (pytest -m gpu): a `.tar` written by the reference's training loop
(run_nerf.py:626-641 -- golden fixture made by oracle/gen_golden.py:gen_checkpoint from the reference's own
NeRF / camera model / CustomAdamOptimizer) is restored by scnerf_amd.create_nerf exactly as the
reference restores it (create_nerf.py:142-172), continues bit-compatibly, and what we save loads back."""
import os
import types

import numpy as np
import pytest
import torch

from scnerf_amd import synthetic as synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
H, W = 20, 30


def make_args(**over):
    spec = synth.camera_spec(H, W, n_cams=4, seed=14, multiplicative=True)
    a = types.SimpleNamespace(
        multires=10, multires_views=4, i_embed=0, use_viewdirs=True, N_importance=8, N_samples=8, netdepth=3,
        netwidth=16, netdepth_fine=3, netwidth_fine=16, netchunk_per_gpu=1024, n_gpus=1, perturb=1.0,
        white_bkgd=False, raw_noise_std=1.0, dataset_type="llff", no_ndc=False, lindisp=False,
        camera_model="pinhole_rot_noise_10k_rayo_rayd", run_without_colmap="none", use_custom_optim=True,
        lrate=5e-4, non_linear_weight_decay=0.1, grid_size=10, ray_o_noise_scale=spec["ray_o_noise_scale"],
        ray_d_noise_scale=spec["ray_d_noise_scale"], extrinsics_noise_scale=spec["extrinsics_noise_scale"],
        intrinsics_noise_scale=spec["intrinsics_noise_scale"], multiplicative_noise=True,
        ft_path=None, basedir=None, expname=None, no_reload=False)
    for k, v in over.items():
        setattr(a, k, v)
    return a, spec


def build(**over):
    from scnerf_amd.create_nerf import create_nerf
    args, spec = make_args(**over)
    return create_nerf(args, float(spec["K_init"][0, 0]), spec["poses"].numpy(), H, W, mode="train", device="cuda")


def trainable(grad_vars):
    return [p for p in grad_vars if p.requires_grad]


def test_reference_checkpoint_restores_and_continues():
    after = np.load(os.path.join(GOLD, "ref_ckpt_after.npz"))
    ckpt = torch.load(os.path.join(GOLD, "ref_ckpt.tar"), map_location="cpu")
    kw_train, kw_test, start, grad_vars, optimizer, cm = build(ft_path=os.path.join(GOLD, "ref_ckpt.tar"))
    assert start == 2
    sd = kw_train["network_fn"].state_dict()
    assert list(sd.keys()) == list(ckpt["network_fn_state_dict"].keys())          # 'module.'-prefixed, same order
    for k, v in ckpt["network_fine_state_dict"].items():
        np.testing.assert_array_equal(kw_train["network_fine"].state_dict()[k].cpu().numpy(), v.numpy())
    for k, v in ckpt["camera_model"].items():
        np.testing.assert_array_equal(cm.state_dict()[k].cpu().numpy(), v.numpy())
    # only the per-parameter state is merged (create_nerf.py:160-163): the groups keep the fresh lr and
    # the training loop re-derives the decayed one from global_step (run_nerf.py:617-621)
    assert optimizer.param_groups[0]["lr"] == 5e-4
    from scnerf_amd.optim import decayed_lr
    lr = decayed_lr(5e-4, 250, start)
    assert lr == ckpt["optimizer_state_dict"]["param_groups"][0]["lr"]
    for grp in optimizer.param_groups:
        grp["lr"] = lr

    ps = trainable(grad_vars)
    assert len(ps) == int(after["n_trainable"])
    optimizer.zero_grad()
    for i, p in enumerate(ps):
        p.grad.copy_(torch.from_numpy(after["grad/%d" % i]).cuda())
    optimizer.step()
    for i, p in enumerate(ps):          # third step of the reference run: needs the restored moments AND step counts
        np.testing.assert_allclose(p.detach().cpu().numpy(), after["after/%d" % i], rtol=2e-6, atol=2e-7, err_msg=str(i))


def test_own_checkpoint_round_trip(tmp_path):
    kw, _, _, grad_vars, optimizer, cm = build()
    g = torch.Generator(device="cuda").manual_seed(3)

    def step(opt, params):
        opt.zero_grad()
        for p in trainable(params):
            p.grad.copy_(torch.randn(p.shape, generator=g, device="cuda") * 0.1)
        opt.step()
    step(optimizer, grad_vars)
    step(optimizer, grad_vars)
    os.makedirs(tmp_path / "exp")
    path = str(tmp_path / "exp" / "000002.tar")
    torch.save({"global_step": 2, "network_fn_state_dict": kw["network_fn"].state_dict(),
                "network_fine_state_dict": kw["network_fine"].state_dict(),
                "optimizer_state_dict": optimizer.state_dict(), "camera_model": cm.state_dict()}, path)
    sd = optimizer.state_dict()
    assert sorted(sd["state"].keys()) == [i for i, p in enumerate(grad_vars) if p.requires_grad]
    assert isinstance(sd["state"][0]["step"], int) and sd["state"][0]["exp_avg"].shape == grad_vars[0].shape
    # the torch-format state loads into a stock torch optimizer too
    torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in grad_vars], lr=1e-3).load_state_dict(
        {"state": sd["state"], "param_groups": [dict(torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))]).state_dict()[
            "param_groups"][0], params=list(range(len(grad_vars))))]})

    kw2, _, start, grad_vars2, optimizer2, cm2 = build(basedir=str(tmp_path), expname="exp")    # found by directory scan
    assert start == 2
    g_state = g.get_state()
    step(optimizer, grad_vars)
    g.set_state(g_state)
    step(optimizer2, grad_vars2)
    for a, b in zip(grad_vars, grad_vars2):
        np.testing.assert_array_equal(a.detach().cpu().numpy(), b.detach().cpu().numpy())
    # no_reload leaves a fresh model
    _, _, start3, _, _, _ = build(basedir=str(tmp_path), expname="exp", no_reload=True)
    assert start3 == 0
