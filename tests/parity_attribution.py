"""TEST INFRASTRUCTURE: attributes fine-stage differences between two fp32 implementations of
`render_rays` to the reference algorithm's own discontinuities.

The hierarchical sampler (/root/reference NeRF/render.py:417-460) is discontinuous in the coarse weights at
two places:
  * the search `inds = searchsorted(cdf, u, right=True)` (:444): when a uniform variate sits within rounding
    of a cdf knot, a 1-ulp change of the knot moves the sample to the neighbouring bin;
  * `denom = where(denom < 1e-5, 1, denom)` (:455-456): a bin whose mass crosses 1e-5 switches between
    "interpolate inside the bin" and "snap to its left edge".
Two implementations whose coarse weights differ in the last bit therefore place a few samples differently, and
the rays owning those samples legitimately differ by more than the 1e-4 bar downstream.  `classify` finds, per
ray, whether one of its samples is in that situation, from the cdf / indices BOTH sides expose."""
import json
import os

import numpy as np
import torch

# everything the parity tests measured; tests/conftest.py writes it at the end of the session to
# gpurun_out/parity_r06.json (travels back from the GPU box; the copy committed under profiles/ is this file)
# and to profiles/ in the tree the tests ran in
REPORT = {}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_report():
    """entries of this session replace the same-named ones of the committed report; the others stay"""
    if not REPORT:
        return
    merged = {}
    try:
        with open(os.path.join(ROOT, "profiles", "parity_r06.json")) as f:
            merged = json.load(f)
    except (OSError, ValueError):
        pass
    merged.update(REPORT)
    for d in ("gpurun_out", "profiles"):
        try:
            os.makedirs(os.path.join(ROOT, d), exist_ok=True)
            with open(os.path.join(ROOT, d, "parity_r06.json"), "w") as f:
                json.dump(merged, f, indent=1, sort_keys=True)
        except OSError:
            pass


def _gather(t, idx):
    return torch.gather(t, 1, idx)


def sample_state(cdf, inds, u, bins):
    """Per-sample quantities of sample_pdf's tail (:446-458) recomputed op by op in fp32 from a side's own
    cdf and indices: (denom before the guard, samples)."""
    cdf, u, bins = torch.as_tensor(cdf).float(), torch.as_tensor(u).float(), torch.as_tensor(bins).float()
    inds = torch.as_tensor(inds).long()
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    c0, c1 = _gather(cdf, below), _gather(cdf, above)
    b0, b1 = _gather(bins, below), _gather(bins, above)
    denom_raw = c1 - c0
    denom = torch.where(denom_raw < 1e-5, torch.ones_like(denom_raw), denom_raw)
    samples = b0 + (u - c0) / denom * (b1 - b0)
    return denom_raw, samples


def classify(u, bins, cdf_a, inds_a, cdf_b, inds_b, illcond=1e-3, displaced=1e-5):
    """-> dict of per-ray boolean arrays:
      index   some sample's search index differs between the sides,
      branch  some sample's `denom < 1e-5` guard takes different branches,
      illcond (neither of the above) some sample sits in a bin of mass < `illcond` on either side AND the two
              sides place it more than `displaced` apart in depth: the continuous map amplifies a 1e-8 cdf
              rounding difference by bin_width / mass (>= 1e3 there; depths themselves round at ~5e-7),
      moved   max over the ray's samples of |sample_a - sample_b| (recomputed from each side's cdf / inds)."""
    da, sa = sample_state(cdf_a, inds_a, u, bins)
    db, sb = sample_state(cdf_b, inds_b, u, bins)
    ia, ib = torch.as_tensor(inds_a).long(), torch.as_tensor(inds_b).long()
    index = (ia != ib).any(-1)
    branch = ((da < 1e-5) != (db < 1e-5)).any(-1)
    moved = (sa - sb).abs()
    thin = ((torch.minimum(da, db) < illcond) & (moved > displaced)).any(-1)
    return dict(index=index.numpy(), branch=branch.numpy(), illcond=(thin & ~index & ~branch).numpy(),
                moved=moved.max(-1)[0].numpy())


def sample_state_npp(cdf, above, u, bins, tiny=1e-6):
    """NeRF++'s sampler tail (nerfplusplus/ddp_train_nerf.py:116-130) recomputed op by op from a side's own cdf and
    comparison-count indices: (denom before the `< TINY` guard, samples)."""
    cdf, u, bins = torch.as_tensor(cdf).float(), torch.as_tensor(u).float(), torch.as_tensor(bins).float()
    above = torch.as_tensor(above).long()
    below = torch.clamp(above - 1, min=0)
    c0, c1 = _gather(cdf, below), _gather(cdf, above)
    b0, b1 = _gather(bins, below), _gather(bins, above)
    denom_raw = c1 - c0
    denom = torch.where(denom_raw < tiny, torch.ones_like(denom_raw), denom_raw)
    return denom_raw, b0 + (u - c0) / denom * (b1 - b0 + tiny)


def classify_npp(u, bins, cdf_a, above_a, cdf_b, above_b, illcond=1e-3, displaced=1e-5, tiny=1e-6):
    """`classify` for the NeRF++ sampler: the index is the count of cdf knots <= u (:113), the guard `denom < 1e-6`
    (:126-127).  Same keys."""
    da, sa = sample_state_npp(cdf_a, above_a, u, bins, tiny)
    db, sb = sample_state_npp(cdf_b, above_b, u, bins, tiny)
    ia, ib = torch.as_tensor(above_a).long(), torch.as_tensor(above_b).long()
    index = (ia != ib).any(-1)
    branch = ((da < tiny) != (db < tiny)).any(-1)
    moved = (sa - sb).abs()
    thin = ((torch.minimum(da, db) < illcond) & (moved > displaced)).any(-1)
    return dict(index=index.numpy(), branch=branch.numpy(), illcond=(thin & ~index & ~branch).numpy(),
                moved=moved.max(-1)[0].numpy())


def merge_causes(*cls):
    """a ray is flagged when any of its samplers (foreground, background) flags it"""
    out = {k: np.zeros_like(cls[0][k]) for k in ("index", "branch", "illcond")}
    for c in cls:
        out["index"] |= c["index"]
        out["branch"] |= c["branch"] & ~out["index"]
    for c in cls:
        out["illcond"] |= c["illcond"]
    out["illcond"] &= ~(out["index"] | out["branch"])
    out["moved"] = np.maximum.reduce([c["moved"] for c in cls])
    return out


def per_ray_error(got, ref, relative=False):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = ref.detach().cpu().double().numpy() if torch.is_tensor(ref) else np.asarray(ref, np.float64)
    e = np.abs(got - ref)
    if relative:
        e = e / np.maximum(np.abs(ref), 1.0)
    return e.reshape(ref.shape[0], -1).max(1)


def gpu_sampling_state(ops, host_linspace, rays, net_c, t_rand, u, noise_c, sc, lindisp=False, white_bkgd=False):
    """The GPU path's own coarse stage re-run kernel by kernel (same kernels, deterministic: bit-identical to
    what RenderRaysFunction computes inside) so that its cdf and search indices can be looked at:
    -> dict(z_c, w_c, rgb0, inds, cdf, z_s, z_f)."""
    with torch.no_grad():
        dev = rays.device
        z_c, pts_c = ops.coarse_sample(rays, host_linspace(sc, dev), t_rand, lindisp)
        flat = net_c.flat_parameters()
        # the same arithmetic the training forward under test used (ops.mlp_arithmetic): the coarse weights decide
        # where the fine samples go
        planes = ops.pack_for_arithmetic(flat, True)
        save = ops.save_workspace(rays.shape[0] * sc, dev) if planes is not None else None
        raw_c = ops.mlp_fwd(pts_c, rays[:, 8:11], sc, ops.pack_weights(flat, "fwd"), save, planes=planes)
        rgb0, _, _, w_c, _ = ops.composite_fwd(raw_c.view(rays.shape[0], sc, 4), z_c, rays, noise_c, white_bkgd)
        z_f, _, z_s, _, inds, cdf = ops.fine_sample(rays, z_c, w_c, u, True, True)
    return dict(z_c=z_c, w_c=w_c, rgb0=rgb0, inds=inds, cdf=cdf, z_s=z_s, z_f=z_f)


def summary(err, cls, bar=1e-4):
    """counts for the report: rays over the bar, and how many of those carry which cause"""
    over = err > bar
    cause = cls["index"] | cls["branch"] | cls["illcond"]
    return dict(rays=int(err.size), over_bar=int(over.sum()), over_bar_index=int((over & cls["index"]).sum()),
                over_bar_branch=int((over & cls["branch"] & ~cls["index"]).sum()),
                over_bar_illcond=int((over & cls["illcond"]).sum()), over_bar_unexplained=int((over & ~cause).sum()),
                rays_with_moved_samples=int(cause.sum()), median=float(np.median(err)),
                p999=float(np.quantile(err, 0.999)), max=float(err.max()),
                max_among_clean_rays=float(err[~cause].max()) if (~cause).any() else 0.0)
