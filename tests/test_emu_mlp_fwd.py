"""Fused MLP forward (HIP source under the CPU SIMT interpreter) vs the oracle network."""
import numpy as np
import pytest
import torch

from oracle import scnerf_oracle as O
from scnerf_amd import mlp_layout as ML
from scnerf_amd import synthetic as synth
from tests.emu import harness as H
from tests.emu_mlp_util import network_params, pack_forward, save_views, oracle_activations

pytestmark = pytest.mark.emu


PDS = [3, 4]


@pytest.mark.parametrize("pd", PDS)
def test_layout_constants_match_kernel(pd):
    lay = ML.layout(pd)
    out = np.zeros(32, np.int32)
    H.call("scnerf_mlp_layout_info", pd, out, 32)
    exp = [lay.fwd_stream, lay.fwd_bias, lay.fwd_bias_f, lay.fwd_bias_v, lay.fwd_bias_rgb, lay.fwd_alpha_w,
           lay.fwd_alpha_b, lay.fwd_total, lay.bwd_stream, lay.bwd_alpha_w, lay.bwd_total,
           lay.save_floats_per_sample, ML.GRAD_FLOATS_PER_SAMPLE]
    assert out[:len(exp)].tolist() == exp
    so, _ = ML.section_offsets(lay.save_sections, 128)       # offsets per padded sample
    go, _ = ML.section_offsets(ML.GRAD_SECTIONS, 128)
    assert out[13:19].tolist() == [so[k] // 128 for k in ("feat", "hv", "epts", "eviews")] + [go[k] // 128 for k in ("dfeat", "dzv")]
    assert int(out[19]) == ML.MASK_WORDS_PER_SAMPLE
    assert int(out[20]) == lay.n_params == {3: 595844, 4: 606596}[pd] == H.lib().scnerf_nerf_param_count(pd)
    assert int(out[21]) == lay.e_width
    for P in (1, 128, 4097):
        assert H.lib().scnerf_mlp_save_floats(pd, P) == lay.save_floats(P)


@pytest.mark.parametrize("pd", PDS)
def test_forward_index_is_a_permutation_of_the_weights(pd):
    lay = ML.layout(pd)
    idx = lay.forward_index()
    used = idx[idx >= 0]
    # every parameter is referenced exactly once by the forward buffer
    cnt = np.bincount(used, minlength=lay.n_params)
    assert cnt.min() == 1 and cnt.max() == 1
    bidx = lay.backward_index()
    cntb = np.bincount(bidx[bidx >= 0], minlength=lay.n_params)
    po = lay.param_offsets
    for name, shape in lay.param_shapes:
        sl = slice(po[name], po[name] + int(np.prod(shape)))
        if name.endswith(".weight"):
            # alpha_linear.weight appears once in the dgrad stream too (VALU table)
            assert cntb[sl].min() == 1 and cntb[sl].max() == 1, name
        else:
            assert cntb[sl].max() == 0, name


@pytest.mark.parametrize("pd", PDS)
@pytest.mark.parametrize("n_rays,spr,save", [(5, 32, True), (3, 64, False)])
def test_mlp_forward_matches_oracle(n_rays, spr, save, pd):
    """pd = 3: the SCNeRF network vs the NeRF oracle; pd = 4: NeRF++'s background net (4-D points) --
    the same oracle formulas apply, its parameters being the NeRF layers under other names."""
    lay = ML.layout(pd)
    IN = lay.in_pts
    p = network_params(0 if pd == 3 else 777, pd)
    wpk = pack_forward(p, pd)
    P = n_rays * spr           # 160: one full workgroup + a ragged one;  192: 1.5 workgroups
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand(P, pd, generator=g) * 3 - 1.5)
    vd = torch.randn(n_rays, 3, generator=g)
    vd = vd / vd.norm(dim=-1, keepdim=True)
    raw = np.full((P, 4), np.nan, np.float32)
    sv = np.full(lay.save_floats(P), np.nan, np.float32) if save else None
    H.call("scnerf_mlp_fwd", pd, pts.numpy(), vd.numpy(), 3, spr, wpk, raw, sv, P, None)
    ref = O.query_network(p, pts.reshape(n_rays, spr, pd), vd).reshape(P, 4)
    np.testing.assert_allclose(raw, ref.numpy(), rtol=2e-5, atol=2e-5)
    if save:
        vps = vd[:, None, :].expand(n_rays, spr, 3).reshape(P, 3)
        oa = oracle_activations(p, pts, vps)
        s = save_views(sv, P, pd)
        assert s["epts"].shape[1] == lay.e_width
        np.testing.assert_allclose(s["epts"][:, :IN], oa["e"].numpy(), rtol=0, atol=2e-6)
        assert np.all(s["epts"][:, IN:] == 0)
        np.testing.assert_allclose(s["eviews"][:, :27], oa["ev"].numpy(), rtol=0, atol=2e-6)
        assert np.all(s["eviews"][:, 27:] == 0)
        for l in range(8):
            np.testing.assert_allclose(s["act%d" % l], oa["acts"][l].numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(s["feat"], oa["feat"].numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(s["hv"], oa["hv"].numpy(), rtol=2e-5, atol=2e-5)
        # lane-native ReLU bit masks: element i = 16 t + r of lane (m, h) <-> feature feat_of(t, r, h),
        # stored in word i >> 5 at bit 31 - (i & 31)
        for sec, ref, ntile in [(l, s["act%d" % l], 8) for l in range(8)] + [(8, s["hv"], 4)]:
            for p_ in (0, 31, 77, P - 1):
                wt, m = divmod(p_, 32)
                for hh in (0, 1):
                    words = s["mask"][sec, wt, m + 32 * hh]
                    for t_ in range(ntile):
                        for r in range(16):
                            i = 16 * t_ + r
                            bit = (int(words[i >> 5]) >> (31 - (i & 31))) & 1
                            assert bit == int(ref[p_, ML.feat_of(t_, r, hh)] > 0)
