"""A size-independent property at the reference's largest single chunk (`--chunk 32768` rays, config_argparse.py:34):
rays are rendered independently (render.py:186-300), so a training step over 32 768 rays x (64 + 128) -- 6.3 M
samples, workspaces of tens of GB, offsets beyond 2^31 floats -- must give every ray EXACTLY what a step over a
4096-ray slice of the same inputs gives it (outputs and ray gradients bit for bit: the resident kernels scale per
sample, not per batch), and the weight gradients of the whole must be the sum of the slices' (fixed-order sums of
another chunking: to rounding)."""
import pytest
import torch

from scnerf_amd import synthetic as synth

pytestmark = pytest.mark.gpu

N, SLICE, SC, SF = 32768, 4096, 64, 128
KEYS = ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "raw", "z_std")


def _step(render, query, nets, rays, rnd):
    rays = rays.clone().requires_grad_(True)
    ret = render.render_rays(rays, nets[0], query, SC, retraw=True, perturb=1.0, N_importance=SF, network_fine=nets[1],
                             raw_noise_std=1.0, _randoms=rnd)
    loss = (ret["rgb_map"] ** 2).sum() + (ret["rgb0"] ** 2).sum() + ret["disp_map"].sum() + ret["acc0"].sum()
    params = [list(net.parameters()) for net in nets]
    got = torch.autograd.grad(loss, [rays] + params[0] + params[1])
    n0 = len(params[0])
    grads = [torch.cat([g.reshape(-1) for g in got[1:1 + n0]]), torch.cat([g.reshape(-1) for g in got[1 + n0:]])]
    return {k: ret[k].detach() for k in KEYS}, got[0].detach(), grads


def test_every_ray_of_a_32768_ray_step_is_what_its_4096_ray_slice_gives():
    from scnerf_amd import create_nerf, ops, render, run_nerf_helpers as H
    ops.check_layout()
    free, _ = torch.cuda.mem_get_info()
    if free < 200 * 2 ** 30:
        pytest.skip("needs ~150 GB of free HBM")

    def make(seed):
        net = H.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        net.load_state_dict(synth.network_params(seed=seed))
        return net.cuda()

    nets = (make(0), make(1))
    e, _ = H.get_embedder(10, 0)
    ed, _ = H.get_embedder(4, 0)
    query = create_nerf.FusedNetworkQuery(e, ed)
    rays = synth.ray_batch(N, seed=5).cuda()
    rnd = {k: v.cuda() for k, v in synth.render_randoms(N, SC, SF, seed=6).items()}
    whole, d_rays, g_whole = _step(render, query, nets, rays, rnd)
    assert all(bool(torch.isfinite(whole[k]).all()) for k in KEYS) and bool(torch.isfinite(d_rays).all())
    g_sum = [torch.zeros_like(g, dtype=torch.float64) for g in g_whole]
    for s in range(0, N, SLICE):
        sl = slice(s, s + SLICE)
        part, d_part, g_part = _step(render, query, nets, rays[sl].contiguous(), {k: v[sl].contiguous() for k, v in rnd.items()})
        if s in (0, N - SLICE, N // 2):                       # first, middle, last slice: every output bit for bit
            for k in KEYS:
                assert torch.equal(part[k], whole[k][sl]), (k, s)
            assert torch.equal(d_part, d_rays[sl]), ("d_rays", s)
        for a, g in zip(g_sum, g_part):
            a += g.double()
    for name, a, g in zip(("coarse", "fine"), g_sum, g_whole):
        err = float((a - g.double()).abs().max() / a.abs().max())
        assert err <= 2e-6, (name, err)
