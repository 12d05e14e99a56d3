"""Helpers shared by the emulated-MLP tests (numpy side of packing / workspaces)."""
import numpy as np
import torch

from scnerf_amd import mlp_layout as ML
from tests.emu import harness as H


def flat_params(p):
    return np.concatenate([p[name].detach().numpy().reshape(-1) for name, _ in ML.PARAM_SHAPES]).astype(np.float32)


def pack_forward(p):
    src = flat_params(p)
    assert src.shape[0] == ML.N_PARAMS
    idx = ML.forward_index()
    dst = np.zeros(idx.shape[0], np.float32)
    H.call("scnerf_gather_f32", src, idx, dst, idx.shape[0], None)
    ref = np.where(idx >= 0, src[np.maximum(idx, 0)], 0).astype(np.float32)
    np.testing.assert_array_equal(dst, ref)
    return dst


def pack_backward(p):
    src = flat_params(p)
    idx = ML.backward_index()
    dst = np.zeros(idx.shape[0], np.float32)
    H.call("scnerf_gather_f32", src, idx, dst, idx.shape[0], None)
    return dst


def save_views(save, P):
    """row-major [P, width] views of every section of the activation workspace (+ the bit masks)."""
    off, total = ML.section_offsets(ML.SAVE_SECTIONS, P)
    assert save.shape[0] == ML.save_floats(P)
    Pp = ML.padded_samples(P)
    out = {"mask": save[total:].view(np.uint32).reshape(9, Pp // 32, 64, 4)}
    for name, w in ML.SAVE_SECTIONS:
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
