"""render_rays with an OPAQUE network callable (reference NeRF/render.py:186-300, create_nerf.py:67-69; SURVEY 8b: "the
fast path must introspect them safely and otherwise fall back to calling it exactly as the reference does"): a network
shape the fused kernels do not cover, and a caller's own closure.  Only the network evaluation leaves the HIP kernels; the
stratified depths, compositing (forward and backward), cdf, wave-ballot search, inverse cdf and merge stay on them, so the
sample indices are the fused path's.  Checked against the CPU oracle with the same injected randoms."""
import numpy as np
import pytest
import torch

from oracle import scnerf_oracle as O
from scnerf_amd import synthetic as synth


def _small_net(dev, seed, D=4, W=64, skips=(2,), multires=6, multires_views=3):
    from scnerf_amd.run_nerf_helpers import NeRF, get_embedder
    embed, in_ch = get_embedder(multires, 0)
    embeddirs, in_views = get_embedder(multires_views, 0)
    torch.manual_seed(seed)
    net = NeRF(D=D, W=W, input_ch=in_ch, input_ch_views=in_views, output_ch=5, skips=list(skips), use_viewdirs=True)
    with torch.no_grad():
        for prm in net.parameters():                       # (biases away from zero: every term of the chain is exercised)
            if prm.dim() == 1:
                prm.uniform_(-0.1, 0.1)
    assert not net.is_standard()
    return net.to(dev), embed, embeddirs


def _case(dev, n, sc, sf):
    from scnerf_amd.create_nerf import FusedNetworkQuery
    from scnerf_amd.render import render_rays
    skips, mr, mrv = (2,), 6, 3
    net_c, embed, embeddirs = _small_net(dev, 11, skips=skips, multires=mr, multires_views=mrv)
    net_f, _, _ = _small_net(dev, 12, skips=skips, multires=mr, multires_views=mrv)
    query = FusedNetworkQuery(embed, embeddirs)
    assert not query.fused_for(net_c)
    rays = synth.ray_batch(n, seed=1)
    rnd = synth.render_randoms(n, sc, sf, seed=3)
    target = synth.target_rgb(n, seed=2)
    rays_d = rays.clone().to(dev).requires_grad_(True)
    ret = render_rays(rays_d, net_c, query, sc, retraw=True, perturb=1.0, N_importance=sf, network_fine=net_f,
                      raw_noise_std=1.0, _randoms={k: v.to(dev) for k, v in rnd.items()})
    assert list(ret) == ["rgb_map", "disp_map", "acc_map", "raw", "rgb0", "disp0", "acc0", "z_std"]
    loss = torch.mean((ret["rgb_map"] - target.to(dev)) ** 2) + torch.mean((ret["rgb0"] - target.to(dev)) ** 2)
    loss.backward()
    pc = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net_c.state_dict().items()}
    pf = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net_f.state_dict().items()}
    rays_c = rays.detach().clone().requires_grad_(True)
    out = O.render_rays(rays_c, pc, pf, sc, sf, rnd["t_rand"], rnd["u"], rnd["noise_c"], rnd["noise_f"], multires=mr,
                        multires_views=mrv, skips=skips, rowsum="aten")
    ref_loss = torch.mean((out["rgb_map"] - target) ** 2) + torch.mean((out["rgb0"] - target) ** 2)
    ref_loss.backward()
    for name in ("rgb0", "acc0"):                                    # coarse stage: every ray
        np.testing.assert_allclose(ret[name].detach().cpu().numpy(), out[name].detach().numpy(), rtol=0, atol=1e-4, err_msg=name)
    # fine stage: behind the sampler, which is discontinuous in the coarse weights (tests/parity_attribution.py) -- the device's
    # GEMM library rounds them differently from the CPU's, so a sample in a few hundred may land in the neighbouring bin
    for name in ("rgb_map", "acc_map"):
        e = np.abs(ret[name].detach().cpu().numpy() - out[name].detach().numpy()).reshape(n, -1).max(1)
        assert (e <= 1e-4).mean() >= 0.98 and e.max() < 2e-2, (name, (e <= 1e-4).mean(), e.max())
    got, ref = ret["raw"].detach().cpu().numpy(), out["raw"].detach().numpy()
    assert (np.abs(got - ref) <= 1e-4 + 1e-4 * np.abs(ref)).mean() >= 0.998
    np.testing.assert_allclose(float(loss.detach()), float(ref_loss.detach()), rtol=2e-4)
    for net, ref in ((net_c, pc), (net_f, pf)):
        for name, prm in net.named_parameters():
            g, r = prm.grad.cpu().numpy().astype(np.float64), ref[name].grad.numpy().astype(np.float64)
            assert np.linalg.norm(g - r) <= 5e-3 * np.linalg.norm(r) + 1e-9, name            # (sums over all rays: l2)
    g, r = rays_d.grad.cpu().numpy()[:, :6], rays_c.grad.numpy()[:, :6]
    e = np.abs(g - r).max(1) / (np.abs(r).max() + 1e-30)
    assert (e <= 2e-3).mean() >= 0.98, (e <= 2e-3).mean()                                         # (per ray: the moved ones differ)
    # a closure of the caller's own around the STANDARD network: opaque too -- and equal to the fused path's result
    from scnerf_amd.run_nerf_helpers import NeRF, get_embedder
    std = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    std.load_state_dict(synth.network_params(seed=0))
    std = std.to(dev)
    fq = FusedNetworkQuery(get_embedder(10, 0)[0], get_embedder(4, 0)[0])
    small = {k: v[:8].to(dev) for k, v in rnd.items()}
    with torch.no_grad():
        a = render_rays(rays[:8].to(dev), std, fq, sc, perturb=1.0, N_importance=sf, network_fine=std, raw_noise_std=1.0, _randoms=small)
        b = render_rays(rays[:8].to(dev), std, lambda p, v, f: fq(p, v, f), sc, perturb=1.0, N_importance=sf, network_fine=std,
                        raw_noise_std=1.0, _randoms=small)
    for name in ("rgb0", "rgb_map", "disp_map", "acc_map", "z_std"):
        np.testing.assert_allclose(b[name].cpu().numpy(), a[name].cpu().numpy(), rtol=0, atol=2e-5, err_msg=name)


def test_opaque_callable_on_the_simt_interpreter():
    from tests.emu.host_on_emu import emulated_device
    with emulated_device():
        _case("cpu", 12, 16, 24)


@pytest.mark.gpu
def test_opaque_callable_gpu():
    _case("cuda", 300, 64, 128)
