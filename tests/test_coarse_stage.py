"""The coarse stage of render_rays as one launch (scnerf_coarse_stage_fwd: stratified depths in the network kernel's
prologue, compositing in its epilogue) against the three launches it replaces -- every output bit for bit, training
and inference instantiations, jitter / noise / white background on and off, odd ray counts (a workgroup holding one
live ray)."""
import numpy as np
import pytest
import torch

from scnerf_amd import synthetic as synth


def _check(dev, n_list):
    from scnerf_amd import mlp_layout as ML, ops
    from scnerf_amd.functional import host_linspace
    p = synth.network_params(seed=0)
    flat = torch.cat([p[name].reshape(-1) for name, _ in ML.PARAM_SHAPES]).to(dev)
    wf = ops.pack_weights(flat, "fwd")
    for n, jitter, noise_on, wb, lindisp, train in n_list:
        rays = synth.ray_batch(n, seed=5, lindisp=lindisp).to(dev)
        rnd = synth.render_randoms(n, 64, 8, seed=7)
        t_rand = rnd["t_rand"].to(dev) if jitter else None
        noise = rnd["noise_c"].to(dev) if noise_on else None
        t_vals = host_linspace(64, dev)
        save_a = ops.save_workspace(n * 64, dev) if train else None
        save_b = ops.save_workspace(n * 64, dev) if train else None
        if train:
            save_a.zero_(), save_b.zero_()
        z0, pts0 = ops.coarse_sample(rays, t_vals, t_rand, lindisp)
        raw0 = ops.mlp_fwd(pts0, rays[:, 8:11], 64, wf, save_a).view(n, 64, 4)
        rgb0, disp0, acc0, w0, depth0 = ops.composite_fwd(raw0, z0, rays, noise, wb)
        z1, pts1, raw1, rgb1, disp1, acc1, w1, depth1 = ops.coarse_stage_fwd(rays, t_vals, t_rand, lindisp, wf, save_b, noise, wb)
        for name, a, b in (("z", z0, z1), ("pts", pts0, pts1), ("raw", raw0, raw1), ("rgb", rgb0, rgb1), ("disp", disp0, disp1),
                           ("acc", acc0, acc1), ("weights", w0, w1), ("depth", depth0, depth1)):
            assert torch.equal(a, b), (name, n, jitter, noise_on, wb, lindisp, train)
        if train:
            assert torch.equal(save_a.view(torch.int32), save_b.view(torch.int32))      # (mask words are not floats)


CASES_SMALL = [(3, True, True, False, False, True), (2, False, False, True, True, False), (1, True, False, False, False, False)]


def test_coarse_stage_on_the_simt_interpreter():
    from tests.emu.host_on_emu import emulated_device
    with emulated_device():
        _check("cpu", CASES_SMALL)


@pytest.mark.gpu
def test_coarse_stage_gpu():
    _check("cuda", CASES_SMALL + [(4096, True, True, False, False, True), (4097, True, True, True, False, False),
                                  (1025, False, True, False, True, True)])
