"""Fused Adam: oracle pinned to the reference's CustomAdamOptimizer / torch.optim.Adam goldens; the HIP
kernel under the CPU SIMT interpreter; the optimizer classes on the GPU."""
import types

import numpy as np
import pytest
import torch

from oracle import scnerf_oracle as O
from conftest import t

SHAPES = [(37, 5), (64,), (3, 4, 3), (3, 4, 3)]


def run_oracle(g, tag, wd, decay_from):
    ps = [t(g["p0/%d" % i]).clone() for i in range(4)]
    m = [torch.zeros_like(p) for p in ps]
    v = [torch.zeros_like(p) for p in ps]
    lr = 5e-4
    outs = []
    for k in range(4):
        grads = [t(g["grad%d/%d" % (k, i)]) for i in range(4)]
        O.adam_step(ps, grads, m, v, [k + 1] * 4, lr, weight_decay=wd, decay_idx_from=decay_from)
        lr = O.lr_schedule(5e-4, 250, k + 1)
        outs.append([p.clone() for p in ps])
    return outs


@pytest.mark.parametrize("tag,wd,decay_from", [("custom_wd", 0.1, 2), ("custom_nowd", 0.0, 2), ("adam", 0.0, 4)])
def test_oracle_adam_pinned_to_reference(golden, tag, wd, decay_from):
    g = golden("optimizer")
    outs = run_oracle(g, tag, wd, decay_from)
    for k in range(4):
        for i in range(4):
            np.testing.assert_allclose(outs[k][i].numpy(), g["%s/step%d/p%d" % (tag, k, i)], rtol=1e-6, atol=1e-7)


@pytest.mark.emu
@pytest.mark.parametrize("n,wd", [(1000, 0.0), (1003, 0.1), (3, 0.0)])
def test_adam_kernel_emulated(n, wd):
    from tests.emu import harness as H
    rng = np.random.default_rng(n)
    p = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    pt, mt, vt = torch.from_numpy(p.copy()), torch.zeros(n), torch.zeros(n)
    lr = 5e-4
    for k in range(3):
        g = (rng.standard_normal(n) * (k + 1)).astype(np.float32)
        st = H.lib().scnerf_adam_step(H.ptr(p), H.ptr(g), H.ptr(m), H.ptr(v), n, lr, 0.9, 0.999, 1e-8, wd, k + 1, None)
        assert st == 0
        O.adam_step([pt], [torch.from_numpy(g)], [mt], [vt], [k + 1], lr, weight_decay=wd, decay_idx_from=0)
        np.testing.assert_allclose(p, pt.numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(v, vt.numpy(), rtol=2e-6, atol=1e-12)
        lr = O.lr_schedule(5e-4, 250, k + 1)


@pytest.mark.gpu
def test_custom_adam_optimizer_matches_reference_golden(golden):
    from scnerf_amd.optim import CustomAdamOptimizer, FusedAdam, decayed_lr
    g = golden("optimizer")
    args = types.SimpleNamespace(camera_model="pinhole_rot_noise_10k_rayo_rayd")
    for tag, wd in (("custom_wd", 0.1), ("custom_nowd", 0.0), ("adam", None)):
        # the first two tensors live in ONE flat buffer (as a NeRF's parameters do) -> one fused segment
        flat = torch.cat([t(g["p0/0"]).reshape(-1), t(g["p0/1"]).reshape(-1)]).cuda()
        ps = [torch.nn.Parameter(flat[:185].view(37, 5)), torch.nn.Parameter(flat[185:].view(64)),
              torch.nn.Parameter(t(g["p0/2"]).cuda()), torch.nn.Parameter(t(g["p0/3"]).cuda())]
        ps[0].data = flat[:185].view(37, 5)
        ps[1].data = flat[185:].view(64)
        opt = (FusedAdam(ps, lr=5e-4) if wd is None else
               CustomAdamOptimizer(params=ps, lr=5e-4, betas=(0.9, 0.999), weight_decay=wd, H=12, W=16, args=args))
        assert len(opt.segments()) == 3 and opt.segments()[0].n == 185 + 64
        for k in range(4):
            opt.zero_grad()
            for i, pp in enumerate(ps):
                pp.grad.copy_(t(g["grad%d/%d" % (k, i)]).cuda())      # autograd would accumulate here
            opt.step()
            for grp in opt.param_groups:
                grp["lr"] = decayed_lr(5e-4, 250, k + 1)
            for i, pp in enumerate(ps):
                np.testing.assert_allclose(pp.detach().cpu().numpy(), g["%s/step%d/p%d" % (tag, k, i)],
                                           rtol=2e-6, atol=1e-7, err_msg="%s step %d p%d" % (tag, k, i))


@pytest.mark.gpu
def test_fused_adam_trains_nerf_and_respects_frozen_parameters():
    """End to end: render -> loss -> backward -> fused step on both networks' flat buffers; a frozen
    camera tensor (requires_grad_(False), the reference's curriculum) is skipped and keeps its step."""
    from scnerf_amd import synthetic as synth
    from scnerf_amd.create_nerf import FusedNetworkQuery
    from scnerf_amd.optim import FusedAdam
    from scnerf_amd.render import render_rays
    from scnerf_amd.run_nerf_helpers import NeRF, get_embedder

    def make(seed):
        net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        net.load_state_dict(synth.network_params(seed=seed))
        return net.cuda()
    net_c, net_f = make(0), make(1)
    net_c.flat_parameters(), net_f.flat_parameters()
    extra = torch.nn.Parameter(torch.ones(8, device="cuda"))
    frozen = torch.nn.Parameter(torch.ones(8, device="cuda"), requires_grad=False)
    opt = FusedAdam(list(net_c.parameters()) + list(net_f.parameters()) + [frozen, extra], lr=5e-4)
    assert [s.n for s in opt.segments()] == [595844, 595844, 8]
    query = FusedNetworkQuery(get_embedder(10, 0)[0], get_embedder(4, 0)[0])
    rays = synth.ray_batch(256, seed=1).cuda()
    target = synth.target_rgb(256, seed=2).cuda()
    losses = []
    for it in range(6):
        opt.zero_grad()
        ret = render_rays(rays, net_c, query, 64, retraw=True, perturb=0.0, N_importance=128, network_fine=net_f)
        loss = torch.mean((ret["rgb_map"] - target) ** 2) + torch.mean((ret["rgb0"] - target) ** 2) + extra.sum() * 0
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]                         # it learns
    assert float(frozen.sum()) == 8.0
    assert net_c.pts_linears[0].weight.data_ptr() == net_c.flat_parameters().data_ptr()   # still one buffer


@pytest.mark.emu
def test_adam_kernel_decays_an_element_range():
    """scnerf_adam_step_range: weight decay on elements [lo, hi) only, with lo at an odd offset (a tensor boundary
    inside a flat segment need not be 16-byte aligned)."""
    from tests.emu import harness as H
    n, lo, hi, wd, lr = 1003, 617, 1003, 0.1, 5e-4
    rng = np.random.default_rng(7)
    p = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    pt = [torch.from_numpy(p[:lo].copy()), torch.from_numpy(p[lo:].copy())]
    mt, vt = [torch.zeros(lo), torch.zeros(n - lo)], [torch.zeros(lo), torch.zeros(n - lo)]
    for k in range(3):
        g = (rng.standard_normal(n) * (k + 1)).astype(np.float32)
        st = H.lib().scnerf_adam_step_range(H.ptr(p), H.ptr(g), H.ptr(m), H.ptr(v), n, lr, 0.9, 0.999, 1e-8, wd, lo, hi, k + 1, None)
        assert st == 0
        O.adam_step(pt, [torch.from_numpy(g[:lo]), torch.from_numpy(g[lo:])], mt, vt, [k + 1] * 2, lr, weight_decay=wd, decay_idx_from=1)
        np.testing.assert_allclose(p, torch.cat(pt).numpy(), rtol=2e-6, atol=1e-7)
    assert H.lib().scnerf_adam_step_range(H.ptr(p), H.ptr(g), H.ptr(m), H.ptr(v), n, lr, 0.9, 0.999, 1e-8, wd, 5, n + 1, 1, None) != 0


def _demo_curriculum_case(dev):
    """The reference's published demo.sh configuration at its start: --use_custom_optim, --non_linear_weight_decay
    0.1, a `..._rayo_rayd` camera model with every camera tensor still frozen (i < add_ie).  The decayed tail of the
    STEPPED list (create_nerf.py:216-226) is then the fine network's rgb_linear weight and bias -- in the middle of
    that network's flat buffer, at float offset 595 457 (not a multiple of 4).  The optimizer must keep one segment per
    network (no padding inside a network's arena range: the weight-gradient kernels add straight into it), decay
    exactly those two tensors, and match the reference rule applied by the oracle."""
    from scnerf_amd import synthetic as synth
    from scnerf_amd.optim import CustomAdamOptimizer
    from scnerf_amd.run_nerf_helpers import NeRF

    def make(seed):
        net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        net.load_state_dict(synth.network_params(seed=seed))
        return net.to(dev)
    net_c, net_f = make(0), make(1)
    net_c.flat_parameters(), net_f.flat_parameters()
    cam = [torch.nn.Parameter(torch.zeros(s, device=dev), requires_grad=False) for s in ((4,), (17, 9), (12, 16, 3), (12, 16, 3))]
    params = list(net_c.parameters()) + list(net_f.parameters()) + cam
    args = types.SimpleNamespace(camera_model="pinhole_rot_noise_10k_rayo_rayd")
    opt = CustomAdamOptimizer(params=params, lr=5e-4, betas=(0.9, 0.999), weight_decay=0.1, H=12, W=16, args=args)
    opt.zero_grad()                                    # (this is where the unaligned split used to raise)
    segs = opt.segments()
    assert [s.n for s in segs] == [595844, 595844]
    assert segs[0].decay[0] == segs[0].decay[1] and segs[1].decay == (595844 - 3 * 128 - 3, 595844)
    assert net_c.attached_flat_grad() is not None and net_f.attached_flat_grad() is not None
    g = torch.Generator().manual_seed(3)
    active = [p for p in params if p.requires_grad]
    ref_p = [p.detach().cpu().clone() for p in active]
    ref_m = [torch.zeros_like(p) for p in ref_p]
    ref_v = [torch.zeros_like(p) for p in ref_p]
    for k in range(2):
        opt.zero_grad()
        grads = [torch.randn(p.shape, generator=g) * 1e-2 for p in ref_p]
        for p, gr in zip(active, grads):
            p.grad.copy_(gr.to(dev))
        opt.step()
        O.adam_step(ref_p, grads, ref_m, ref_v, [k + 1] * len(ref_p), 5e-4, weight_decay=0.1, decay_idx_from=len(ref_p) - 2)
    for p, r, name in zip(active, ref_p, [n for n, _ in net_c.named_parameters()] + [n for n, _ in net_f.named_parameters()]):
        np.testing.assert_allclose(p.detach().cpu().numpy(), r.numpy(), rtol=3e-6, atol=1e-7, err_msg=name)
    # the curriculum switches the camera tensors on (i == add_ie): the tail moves to ray_o / ray_d, the networks'
    # segments stay whole and keep their moments and step counts
    for c in cam:
        c.requires_grad_(True)
    opt.zero_grad()
    segs = opt.segments()
    assert [s.n for s in segs][:2] == [595844, 595844] and all(s.decay[0] == s.decay[1] for s in segs[:2])
    assert [s.step for s in segs[:2]] == [2, 2] and float(segs[1].exp_avg.abs().sum()) > 0
    assert sum(s.decay[1] - s.decay[0] for s in segs) == 2 * 12 * 16 * 3
    # a tensor frozen again keeps its optimizer state for the day it is unfrozen (checkpoints carry it too)
    for p, gr in zip(cam, [torch.ones(c.shape) for c in cam]):
        p.grad.copy_(gr.to(dev))
    opt.step()
    cam[0].requires_grad_(False)
    opt.zero_grad()
    sd = opt.state_dict()
    idx_cam0 = len(params) - 4
    assert idx_cam0 in sd["state"] and int(float(sd["state"][idx_cam0]["step"])) == 1
    cam[0].requires_grad_(True)
    opt.zero_grad()
    assert [s.step for s in opt.segments() if any(q is cam[0] for q in s.params)][0] == 1


@pytest.mark.emu
def test_demo_sh_curriculum_on_the_simt_interpreter():
    from tests.emu.host_on_emu import emulated_device
    with emulated_device():
        _demo_curriculum_case(torch.device("cpu"))


@pytest.mark.gpu
def test_demo_sh_curriculum_gpu():
    _demo_curriculum_case(torch.device("cuda"))
