"""Projected-ray-distance loss: the oracle restatement and the HIP kernels (under the CPU SIMT
interpreter) against golden vectors of the reference's proj_ray_dist_loss_single
(model/ray_dist_loss.py:22-246): loss, n_match and every input gradient."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import scnerf_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "prd.npz"))
NAMES = ("rays0_o", "rays0_d", "rays1_o", "rays1_d")


def case(tag):
    k = tag + "/"
    i0, i1 = G[k + "idx"]
    d = {n: G[k + n] for n in NAMES + ("kps0", "kps1", "K")}
    d["E2"] = np.ascontiguousarray(G[k + "E"][[i0, i1]])
    d["thr"] = float(G[k + "threshold"])
    return d, (int(i0), int(i1))


def truth64(tag):
    """fp64 evaluation of the (reference-pinned) oracle: the loss is ill-conditioned in fp32 -- nearly
    parallel rays make r^2 - 1 cancel -- so the reference's own fp32 gradients sit ~2e-3 (relative to
    the largest entry) away from the exact ones.  Kernels are held to: as close to the fp32 reference
    as fp32 re-association allows (1e-3), and no further from the exact values than 2x the reference."""
    d, (i0, i1) = case(tag)
    t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=k not in ("kps0", "kps1")) for k, v in d.items()
         if k != "thr"}
    loss, _ = O.prd_loss(t["kps0"], t["kps1"], t["rays0_o"], t["rays0_d"], t["rays1_o"], t["rays1_d"], t["K"],
                         t["E2"], d["thr"])
    loss.backward()
    out = {n: t[n].grad.numpy() for n in NAMES}
    out["K"], out["E"] = t["K"].grad.numpy(), t["E2"].grad.numpy()
    return out


def check_grad(got, ref32, exact, what):
    scale = float(np.abs(ref32).max()) + 1e-30
    assert np.isfinite(got).all(), what
    e_ref = float(np.abs(got - ref32).max())
    assert e_ref <= 1e-3 * scale, "%s vs reference: err %g scale %g" % (what, e_ref, scale)
    e_exact, r_exact = float(np.abs(got - exact).max()), float(np.abs(ref32 - exact).max())
    assert e_exact <= 2.0 * r_exact + 1e-5 * scale, "%s vs fp64: err %g, reference's %g" % (what, e_exact, r_exact)


def close(a, b, tol, what):
    scale = float(np.abs(b).max()) + 1e-30
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, "%s: err %g scale %g" % (what, err, scale)


@pytest.mark.parametrize("tag", ["leaf", "leaf_tight"])
def test_oracle_matches_reference(tag):
    d, (i0, i1) = case(tag)
    t = {k: torch.tensor(v, requires_grad=k not in ("kps0", "kps1")) for k, v in d.items() if k != "thr"}
    loss, nm = O.prd_loss(t["kps0"], t["kps1"], t["rays0_o"], t["rays0_d"], t["rays1_o"], t["rays1_d"], t["K"],
                          t["E2"], d["thr"])
    loss.backward()
    k = tag + "/"
    assert nm == float(G[k + "n_match"])
    assert abs(float(loss.detach()) - float(G[k + "loss"])) <= 1e-6 * abs(float(G[k + "loss"]))
    for n in NAMES:
        close(t[n].grad.numpy(), G[k + "g_" + n], 2e-4, n)
    close(t["K"].grad.numpy(), G[k + "g_K"], 2e-4, "g_K")
    close(t["E2"].grad.numpy(), G[k + "g_E"][[i0, i1]], 2e-4, "g_E")


def test_oracle_eval_mode():
    d, _ = case("leaf")
    t = {k: torch.tensor(v) for k, v in d.items() if k != "thr"}
    loss, nm = O.prd_loss(t["kps0"], t["kps1"], t["rays0_o"], t["rays0_d"], t["rays1_o"], t["rays1_d"], t["K"],
                          t["E2"], d["thr"], eval_mode=True)
    assert nm is None
    assert abs(float(loss) - float(G["eval/loss"])) <= 1e-6 * float(G["eval/loss"])


def emu_fwd(H, d, eval_mode=0, m=None):
    m = d["kps0"].shape[0] if m is None else m
    sums = np.full(6, np.nan, np.float32)
    loss = np.full(1, np.nan, np.float32)
    nm = np.full(1, np.nan, np.float32)
    H.call("scnerf_prd_loss_fwd", d["kps0"], d["kps1"], d["rays0_o"], d["rays0_d"], d["rays1_o"], d["rays1_d"],
           d["K"], d["E2"], ctypes.c_float(1e-10), ctypes.c_float(d["thr"]), 1, eval_mode, m, sums, loss, nm, None)
    return loss[0], nm[0], sums


@pytest.mark.emu
@pytest.mark.parametrize("tag", ["leaf", "leaf_tight"])
def test_emu_kernels_match_reference(tag):
    from tests.emu import harness as H
    d, (i0, i1) = case(tag)
    k = tag + "/"
    m = d["kps0"].shape[0]
    loss, nm, sums = emu_fwd(H, d)
    assert nm == float(G[k + "n_match"])
    assert abs(loss - float(G[k + "loss"])) <= 2e-5 * abs(float(G[k + "loss"]))
    g = [np.full((m, 3), np.nan, np.float32) for _ in range(4)]
    gK = np.full((4, 4), np.nan, np.float32)
    gE = np.full((2, 4, 4), np.nan, np.float32)
    ws = np.zeros(36, np.float32)
    H.call("scnerf_prd_loss_bwd", d["kps0"], d["kps1"], d["rays0_o"], d["rays0_d"], d["rays1_o"], d["rays1_d"],
           d["K"], d["E2"], ctypes.c_float(1e-10), ctypes.c_float(d["thr"]), 1, m, sums, np.ones(1, np.float32),
           g[0], g[1], g[2], g[3], gK, gE, ws, None)
    exact = truth64(tag)
    for n, a in zip(NAMES, g):
        check_grad(a, G[k + "g_" + n], exact[n], n)
    check_grad(gK, G[k + "g_K"], exact["K"], "g_K")
    check_grad(gE, G[k + "g_E"][[i0, i1]], exact["E"], "g_E")


@pytest.mark.emu
def test_emu_eval_mode_and_empty():
    from tests.emu import harness as H
    d, _ = case("leaf")
    loss, _, _ = emu_fwd(H, d, eval_mode=1)
    assert abs(loss - float(G["eval/loss"])) <= 2e-5 * float(G["eval/loss"])
    loss, nm, _ = emu_fwd(H, d, m=0)           # no match survives -> mean of nothing, nan like the reference
    assert np.isnan(loss) and nm == 0
