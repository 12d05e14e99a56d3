"""Helpers shared by the emulated-MLP tests (numpy side of packing / workspaces)."""
import numpy as np
import torch

from scnerf_amd import mlp_layout as ML
from tests.emu import harness as H


NPP_NAMES = {"alpha_linear": "sigma_layers.0", "feature_linear": "base_remap_layers.0",
             "views_linears.0": "rgb_layers.0", "rgb_linear": "rgb_layers.2"}


def npp_to_nerf_names(p, prefix):
    """NeRF++ MLPNet parameters (nerfplusplus/nerf_network.py:86-115) under the NeRF names the layout
    tables use: same tensors, same shapes (base_layers.i.0 = pts_linears.i, sigma = alpha, base_remap =
    feature, rgb_layers.0 / .2 = views_linears.0 / rgb_linear)."""
    out = {}
    for i in range(8):
        for wb in ("weight", "bias"):
            out["pts_linears.%d.%s" % (i, wb)] = p[prefix + "base_layers.%d.0.%s" % (i, wb)]
    for a, b in NPP_NAMES.items():
        for wb in ("weight", "bias"):
            out[a + "." + wb] = p[prefix + b + "." + wb]
    return out


def network_params(seed, pd=3):
    """pd = 3: the synthetic SCNeRF network; pd = 4: the background net of a synthetic NeRF++ model."""
    from scnerf_amd import synthetic as synth
    if pd == 3:
        return synth.network_params(seed=seed)
    return {k: v.clone() for k, v in npp_to_nerf_names(synth.nerfpp_params(seed), "bg_net.").items()}


def flat_params(p, pd=3):
    return np.concatenate([p[name].detach().numpy().reshape(-1) for name, _ in ML.layout(pd).param_shapes]).astype(np.float32)


def pack_forward(p, pd=3):
    src = flat_params(p, pd)
    assert src.shape[0] == ML.layout(pd).n_params
    idx = ML.layout(pd).forward_index()
    dst = np.zeros(idx.shape[0], np.float32)
    H.call("scnerf_gather_f32", src, idx, dst, idx.shape[0], None)
    ref = np.where(idx >= 0, src[np.maximum(idx, 0)], 0).astype(np.float32)
    np.testing.assert_array_equal(dst, ref)
    return dst


def pack_backward(p, pd=3):
    src = flat_params(p, pd)
    idx = ML.layout(pd).backward_index()
    dst = np.zeros(idx.shape[0], np.float32)
    H.call("scnerf_gather_f32", src, idx, dst, idx.shape[0], None)
    return dst


def save_views(save, P, pd=3):
    """row-major [P, width] views of every section of the activation workspace (+ the bit masks)."""
    lay = ML.layout(pd)
    off, total = ML.section_offsets(lay.save_sections, P)
    assert save.shape[0] == lay.save_floats(P)
    Pp = ML.padded_samples(P)
    out = {"mask": save[total:].view(np.uint32).reshape(9, Pp // 32, 64, 4)}
    for name, w in lay.save_sections:
        blk = save[off[name]: off[name] + w * Pp]
        out[name] = ML.untile(blk, w, P) if name in ML.TILED_SECTIONS else blk.reshape(Pp, w)[:P]
    return out


def grad_views(grads, P):
    off, total = ML.section_offsets(ML.GRAD_SECTIONS, P)
    assert grads.shape[0] == ML.grad_floats(P)
    Pp = ML.padded_samples(P)
    return {name: ML.untile(grads[off[name]: off[name] + w * Pp], w, P) for name, w in ML.GRAD_SECTIONS}


def oracle_activations(p, pts, viewdirs_per_sample):
    """Intermediate activations of the reference network (oracle formulas) for checking
    what the training forward saves."""
    from oracle import scnerf_oracle as O
    import torch.nn.functional as F
    e = O.positional_encoding(pts, 10)
    ev = O.positional_encoding(viewdirs_per_sample, 4)
    acts = []
    h = e
    for i in range(8):
        h = F.relu(F.linear(h, p["pts_linears.%d.weight" % i], p["pts_linears.%d.bias" % i]))
        acts.append(h)
        if i == 4:
            h = torch.cat([e, h], -1)
    feat = F.linear(acts[-1], p["feature_linear.weight"], p["feature_linear.bias"])
    hv = F.relu(F.linear(torch.cat([feat, ev], -1), p["views_linears.0.weight"], p["views_linears.0.bias"]))
    return dict(e=e, ev=ev, acts=acts, feat=feat, hv=hv)


def pack_h3(p, pd=3, directions=("fwd", "bwd")):
    """(forward stream, backward stream, scale table) of the resident arithmetic (scnerf_h3_pack on the interpreter)"""
    src = flat_params(p, pd)
    jobs = np.ascontiguousarray(ML.h3_scale_jobs(pd))
    scales = np.zeros(H.lib().scnerf_h3_scale_floats(), np.float32)
    plans = {k: ML.h3_plan(pd, k) for k in directions}
    streams = {k: np.zeros(plans[k][1].shape[0] * 512, np.int16) for k in directions}

    def arg(k, i):
        return plans[k][i] if k in plans else None
    H.call("scnerf_h3_pack", src, jobs, arg("fwd", 0), arg("fwd", 1), plans["fwd"][1].shape[0] if "fwd" in plans else 0,
           arg("bwd", 0), arg("bwd", 1), plans["bwd"][1].shape[0] if "bwd" in plans else 0,
           streams.get("fwd"), streams.get("bwd"), scales, None)
    return streams.get("fwd"), streams.get("bwd"), scales
