"""TEST INFRASTRUCTURE: the weight distributions the parity tests of the resident arithmetic run on.

  "xavier"       scnerf_amd.synthetic.network_params(seed): the reference's initialisation (NeRF/run_nerf_helpers.py:13-21)
  "trained"      tests/golden/trained_nerf.npz: coarse + fine network after 5000 Adam steps on the procedural scene, trained on
                 the MI355X with the exact-fp32 MFMA arithmetic (tools/train_golden_weights.py) -- heavy-tailed rows, dead
                 units, large biases: what the product sees after convergence (reference network: run_nerf_helpers.py:105-128)
  "adversarial"  xavier with three hand-made layers aimed at the per-sample scale bound of csrc/mlp_h3.h:18-26
                   pts_linears.1: every weight x 2^e, e uniform in -20 .. 4 (one layer spanning 2^-20 .. 2^4)
                   pts_linears.2: rows 2k and 2k+1 identical (their activations are equal bit for bit)
                   pts_linears.3: row 7 = (+c, -c) on those pairs with a 1-norm 2^10 x the other rows': its pre-activation is
                                  exactly its bias while it inflates the layer's row-1-norm bound A -- every sample's scale for
                                  this layer sits ten octaves below where its values would put it
"""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_nerf.npz")
KINDS = ("xavier", "trained", "adversarial")
_cache = {}


def have_trained():
    return os.path.isfile(GOLDEN)


def weights(kind, seed=0, which="fine"):
    """-> state dict (name -> fp32 CPU tensor) of the standard SCNeRF network (pd = 3)"""
    from scnerf_amd import synthetic as synth
    if kind == "xavier":
        return synth.network_params(seed=seed)
    if kind == "trained":
        if "npz" not in _cache:
            _cache["npz"] = dict(np.load(GOLDEN))
        pre = which + "/"
        return {k[len(pre):]: torch.from_numpy(v.copy()) for k, v in _cache["npz"].items() if k.startswith(pre)}
    if kind == "adversarial":
        p = {k: v.clone() for k, v in synth.network_params(seed=seed).items()}
        g = torch.Generator().manual_seed(4242 + seed)
        w1 = p["pts_linears.1.weight"]
        p["pts_linears.1.weight"] = w1 * torch.exp2(torch.randint(-20, 5, w1.shape, generator=g).float())
        w2, b2 = p["pts_linears.2.weight"], p["pts_linears.2.bias"]
        w2[1::2] = w2[0::2]
        b2[1::2] = b2[0::2]
        w3 = p["pts_linears.3.weight"]
        others = float(w3.abs().sum(1).median())
        c = others * 1024.0 / 256.0
        w3[7, 0::2] = c
        w3[7, 1::2] = -c
        return p
    raise ValueError(kind)


def scale_margins(ops, planes, save, P, pts):
    """log2 of (largest |value| of a sample's layer output) x (the power of two S the kernel cut that output at), min and max
    over the samples, per trunk layer -- S re-derived on the host from the kernel's own scale table (row-1-norm bound A,
    largest bias B: csrc/mlp_h3.h kBoundA / kBoundB) and the saved activations, by the kernel's rule
    (mlp_fwd_h3_kernel.h: trunk_layer): S = scale_for(max(A max|x_in| + B, floor)).  fp32 grade needs the product in
    [2^-3, 2^13) (mlp_h3.h:23-25)."""
    from scnerf_amd import mlp_layout as ML
    lay = ML.layout(3)
    Pp = ML.padded_samples(P)
    off, _ = ML.section_offsets(lay.save_sections, P)
    sc = planes.scales.view(-1, 8).double()

    def rows(name, width=256):
        blk = save[off[name]: off[name] + width * Pp]
        return blk.view(Pp // 32, width // 32, 4, 2, 32, 4).permute(0, 4, 1, 2, 3, 5).reshape(Pp, width)[:P]

    def scale_for(bound):
        e = torch.floor(torch.log2(bound.clamp_min(2.0 ** -114)))
        return torch.exp2(12.0 - e)
    m_e = pts.double().abs().max(1)[0].clamp_min(1.0)
    out = {}
    am = m_e
    for l in range(8):
        A, B = sc[l, 2], sc[l, 3]
        a_in = am if l != 5 else torch.maximum(am, m_e)            # the skip layer's input includes the encoded point
        bound = A * a_in + B
        if l == 4:
            bound = torch.maximum(bound, m_e)                      # layer 4's output shares its scale with the encoded point
        S = scale_for(bound)
        z = rows("act%d" % l).double()
        zmax = z.abs().max(1)[0]
        live = zmax > 0
        m = torch.log2((zmax * S)[live])
        out["layer_%d" % l] = {"log2_min": float(m.min()), "log2_max": float(m.max()), "samples_all_zero": int((~live).sum())}
        am = zmax
    return out
