"""Forward-only calls take their packed weights from a version-keyed cache on the network (ops.inference_packs; SURVEY
section 8b, reference NeRF/render.py:143-183: render_path renders one image in ~24 chunks with the same weights): packed
once per weight version, re-packed after anything that bumps a version counter, never used by a training call."""
import pytest
import torch

from scnerf_amd import synthetic as synth


def _world(dev):
    from scnerf_amd.create_nerf import FusedNetworkQuery
    from scnerf_amd.run_nerf_helpers import NeRF, get_embedder

    def make(seed):
        net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        net.load_state_dict(synth.network_params(seed=seed))
        return net.to(dev)
    return make(0), make(1), FusedNetworkQuery(get_embedder(10, 0)[0], get_embedder(4, 0)[0])


def _check(dev, n):
    from scnerf_amd import ops
    from scnerf_amd.render import render_rays
    net_c, net_f, query = _world(dev)
    rays = synth.ray_batch(n, seed=3).to(dev)

    def infer():
        with torch.no_grad():
            return render_rays(rays, net_c, query, 64, retraw=False, perturb=0.0, N_importance=8, network_fine=net_f,
                               raw_noise_std=0.0)["rgb_map"].clone()
    assert ops.weight_pack_cache() is True
    st = ops.PACK_CACHE_STATS
    st["hits"] = st["packs"] = 0
    # a call under torch.no_grad() is forward-only even though the parameters require grad (ctx.needs_input_grad alone says
    # otherwise): no activation workspace is allocated, the inference instantiation runs
    calls = []
    real = ops.save_workspace
    ops.save_workspace = lambda *a_, **k_: calls.append(a_) or real(*a_, **k_)
    try:
        a = infer()
    finally:
        ops.save_workspace = real
    assert calls == []
    assert (st["packs"], st["hits"]) == (2, 0)              # coarse + fine network, once each
    b = infer()
    c = infer()
    assert (st["packs"], st["hits"]) == (2, 4) and torch.equal(a, b) and torch.equal(a, c)
    # an in-place update (what every optimizer does) bumps the parameter's version: re-packed, and the result is what a run
    # without the cache gives
    with torch.no_grad():
        net_f.rgb_linear.bias.add_(0.25)
    d = infer()
    assert (st["packs"], st["hits"]) == (3, 5)              # the fine network again; the coarse one was a hit
    assert not torch.equal(a, d)
    ops.weight_pack_cache(False)
    try:
        e = infer()
    finally:
        ops.weight_pack_cache(True)
    assert torch.equal(d, e)
    # load_state_dict copies in place: a new version too
    net_f.load_state_dict(synth.network_params(seed=1))
    assert torch.equal(infer(), a)
    # a training call packs for itself and leaves the cache alone
    packs = st["packs"]
    ret = render_rays(rays, net_c, query, 64, retraw=False, perturb=0.0, N_importance=8, network_fine=net_f, raw_noise_std=0.0)
    ret["rgb_map"].sum().backward()
    assert st["packs"] == packs
    # a write that by-passes version counting is invisible -- forget_packs is the documented way out
    net_f.rgb_linear.bias.data.add_(0.25)
    assert torch.equal(infer(), a)
    ops.forget_packs(net_f)
    assert torch.equal(infer(), d)


def test_pack_cache_on_the_simt_interpreter():
    from tests.emu.host_on_emu import emulated_device
    with emulated_device():
        _check("cpu", 2)


@pytest.mark.gpu
def test_pack_cache_gpu():
    _check("cuda", 700)
