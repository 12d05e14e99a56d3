"""Typed wrappers over the C ABI (include/scnerf_hip.h): torch CUDA(ROCm) tensors in,
kernels enqueued on torch's current stream.  No computation happens in Python and there
is no CPU fallback: a missing library or a CPU tensor raises."""
from __future__ import annotations

from typing import Optional

import numpy as np
import os

import torch

from . import _capi
from . import mlp_layout as ML

Tensor = torch.Tensor


class _Profile:
    """HIP-event timers around the heavy launches (off unless bench.py switches it on).  Events are
    recorded on torch's current stream, which is the stream the kernels are enqueued on."""

    def __init__(self):
        self.enabled = False
        self.records = {}

    def reset(self, enabled=False):
        self.records = {}
        self.enabled = enabled

    class _Region:
        def __init__(self, prof, name, flop, group):
            self.prof, self.name, self.flop, self.group = prof, name, flop, group

        def __enter__(self):
            if self.prof.enabled:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e1 = torch.cuda.Event(enable_timing=True)
                self.e0.record()
            return self

        def __exit__(self, *exc):
            if self.prof.enabled:
                self.e1.record()
                self.prof.records.setdefault(self.name, []).append((self.e0, self.e1, self.flop, self.group))
            return False

    def region(self, name, flop=0, group=False):
        return _Profile._Region(self, name, flop, group)

    def raw_pair(self, name, flop=0):
        """two events the C side records itself (their raw handles exist once recorded here)"""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        self.records.setdefault(name, []).append((e0, e1, flop, False))
        return e0, e1

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _, _ in recs]
            out[name] = {"launches": len(recs), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms),
                         "flop_per_launch": sum(r[2] for r in recs) / len(recs), "group": recs[0][3]}
        return out


PROFILE = _Profile()


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return _capi.current_stream()


def _chk(t: Tensor, dtype, name):
    if not _capi.on_device(t):
        raise RuntimeError("%s must live on the GPU (scnerf_amd has no CPU path)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def _f(t, name):
    return _chk(t, torch.float32, name)


def searchsorted(a: Tensor, v: Tensor, side: str = "right") -> Tensor:
    """Batched search with broadcast rows, int64 result (torchsearchsorted.searchsorted
    semantics, NeRF/torchsearchsorted/src/torchsearchsorted/searchsorted.py:20-53)."""
    _f(a, "a"), _f(v, "v")
    nrow = max(a.shape[0], v.shape[0])
    out = torch.empty((nrow, v.shape[1]), dtype=torch.int64, device=a.device)
    st = _capi.load().scnerf_searchsorted(_p(a), _p(v), _p(out), nrow, a.shape[0], v.shape[0],
                                          a.shape[1], v.shape[1], int(side == "left"), _stream())
    _capi.check(st, "scnerf_searchsorted")
    return out


def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor, want_inds=False, want_cdf=False):
    _f(bins, "bins"), _f(weights, "weights"), _f(u, "u")
    n, nb = bins.shape
    ns = u.shape[-1]
    stride = ns if u.dim() == 2 else 0
    samples = torch.empty((n, ns), dtype=torch.float32, device=bins.device)
    inds = torch.empty((n, ns), dtype=torch.int64, device=bins.device) if want_inds else None
    cdf = torch.empty((n, nb), dtype=torch.float32, device=bins.device) if want_cdf else None
    st = _capi.load().scnerf_sample_pdf(_p(bins), _p(weights), _p(u), stride, _p(samples), _p(inds),
                                        _p(cdf), n, nb, ns, _stream())
    _capi.check(st, "scnerf_sample_pdf")
    return samples, inds, cdf


_random_state = {"shard": None, "draw": None}
_MASK64 = 0xFFFFFFFFFFFFFFFF


def random_shard(shard: Optional[int] = None) -> int:
    """Which slice of the Philox counter space this PROCESS draws from.  Ray-parallel ranks share the seed and make the
    same sequence of calls: without it ray i of every rank would get the same jitter.  Default: torch.distributed's rank
    once a process group is up (else $RANK, else 0); `random_shard(k)` pins it."""
    if shard is not None:
        _random_state["shard"] = int(shard) & 0xFFFFFFFF
    if _random_state["shard"] is not None:
        return _random_state["shard"]
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return int(dist.get_rank())
    except Exception:
        pass
    return int(os.environ.get("RANK", "0") or 0)


def next_random_call(device) -> tuple:
    """-> (seed, call) of the next render_randoms launch.  There is no hidden counter: `call` is one 63-bit draw from
    torch's DEFAULT (CPU) generator (1.4 us on the host, no device work), so torch.manual_seed(s) -- at the start or in the
    middle of a process -- reproduces everything that follows, as it does for the reference's torch.rand calls; the
    process's shard (random_shard) is folded in so that ranks draw apart.  `seed` is torch.initial_seed(); where the device
    generator was seeded on its own (torch.cuda.manual_seed, which the reference's on-device torch.rand would honour) its
    seed is mixed in."""
    seed = torch.initial_seed() & _MASK64
    if torch.device(device).type == "cuda":
        cs = torch.cuda.initial_seed() & _MASK64
        if cs != seed:
            seed = (seed * 0x9E3779B97F4A7C15 + cs) & _MASK64
    if _random_state["draw"] is None:
        _random_state["draw"] = torch.empty((), dtype=torch.int64, device="cpu")
    call = int(_random_state["draw"].random_()) ^ ((random_shard() * 0x9E3779B97F4A7C15) & _MASK64)
    return seed, call


def render_randoms(n: int, n_samples: int, n_importance: int, raw_noise_std: float, device, want_t_rand=True, want_u=True):
    """The draws of one render_rays call in ONE launch (csrc/randoms.hip; reference NeRF/render.py:252-257, :329-330,
    :425-429): -> (t_rand [n, S] | None, u [n, N_importance] | None, noise_c [n, S] | None, noise_f [n, S + N_importance] |
    None), the noises already multiplied by raw_noise_std.  Philox4x32-10, key and call id from `next_random_call`:
    reproducible from torch.manual_seed, different on every rank."""
    S, F = int(n_samples), int(n_importance)
    sizes = [n * S if want_t_rand else 0, n * F if (want_u and F > 0) else 0,
             n * S if raw_noise_std > 0 else 0, n * (S + F) if (raw_noise_std > 0 and F > 0) else 0]
    padded = [(z + 3) // 4 * 4 for z in sizes]
    buf = torch.empty(sum(padded), dtype=torch.float32, device=device)
    outs, o = [], 0
    for z, pz in zip(sizes, padded):
        outs.append(buf[o:o + z] if z else None)
        o += pz
    if sum(sizes) == 0:
        return (None, None, None, None)
    seed, call = next_random_call(device)
    st = _capi.load().scnerf_render_randoms(seed, call, _p(outs[0]), sizes[0], _p(outs[1]), sizes[1], _p(outs[2]), sizes[2],
                                            _p(outs[3]), sizes[3], float(max(raw_noise_std, 0.0)), _stream())
    _capi.check(st, "scnerf_render_randoms")
    shapes = [(n, S), (n, F), (n, S), (n, S + F)]
    return tuple(None if t is None else t.view(sh) for t, sh in zip(outs, shapes))


def coarse_sample(rays: Tensor, t_vals: Tensor, t_rand: Optional[Tensor], lindisp: bool):
    _f(rays, "rays"), _f(t_vals, "t_vals")
    n, s = rays.shape[0], t_vals.shape[0]
    if t_rand is not None:
        _f(t_rand, "t_rand")
    z = torch.empty((n, s), dtype=torch.float32, device=rays.device)
    pts = torch.empty((n, s, 3), dtype=torch.float32, device=rays.device)
    st = _capi.load().scnerf_coarse_sample(_p(rays), rays.shape[1], _p(t_vals), _p(t_rand), _p(z),
                                           _p(pts), n, s, int(bool(lindisp)), _stream())
    _capi.check(st, "scnerf_coarse_sample")
    return z, pts


def fine_sample(rays: Tensor, z_c: Tensor, w_c: Tensor, u: Tensor, want_inds=False, want_cdf=False):
    _f(rays, "rays"), _f(z_c, "z_c"), _f(w_c, "w_c"), _f(u, "u")
    n, sc = z_c.shape
    sf = u.shape[-1]
    stride = sf if u.dim() == 2 else 0
    dev = rays.device
    z_f = torch.empty((n, sc + sf), dtype=torch.float32, device=dev)
    pts_f = torch.empty((n, sc + sf, 3), dtype=torch.float32, device=dev)
    z_s = torch.empty((n, sf), dtype=torch.float32, device=dev)
    z_std = torch.empty((n,), dtype=torch.float32, device=dev)
    inds = torch.empty((n, sf), dtype=torch.int64, device=dev) if want_inds else None
    cdf = torch.empty((n, sc - 1), dtype=torch.float32, device=dev) if want_cdf else None
    st = _capi.load().scnerf_fine_sample(_p(rays), rays.shape[1], _p(z_c), _p(w_c), _p(u), stride,
                                         _p(z_f), _p(pts_f), _p(z_s), _p(z_std), _p(inds), _p(cdf),
                                         n, sc, sf, _stream())
    _capi.check(st, "scnerf_fine_sample")
    return z_f, pts_f, z_s, z_std, inds, cdf


_index_cache = {}


def _device_index(kind: str, device, pd: int = 3, remap=None) -> Tensor:
    """int32 gather table (flat parameter buffer -> streaming order) of network variant `pd` on the
    device.  `remap` = (key, int64 array [n_params]) translates the table's canonical parameter offsets
    (mlp_layout.Layout order) into the offsets of a module that registers the same tensors in another
    order (NeRF++'s MLPNet)."""
    key = (kind, str(device), pd, None if remap is None else remap[0])
    if key not in _index_cache:
        lay = ML.layout(pd)
        idx = lay.forward_index() if kind == "fwd" else lay.backward_index()
        if remap is not None:
            table = np.asarray(remap[1], dtype=np.int64)
            idx = np.where(idx >= 0, table[np.maximum(idx, 0)], -1).astype(np.int32)
        _index_cache[key] = torch.from_numpy(idx).to(device)
    return _index_cache[key]


def check_layout():
    import ctypes
    lib = _capi.load()
    for pd in (3, 4):
        lay = ML.layout(pd)
        out = np.zeros(32, np.int32)
        st = lib.scnerf_mlp_layout_info(pd, out.ctypes.data_as(ctypes.c_void_p), 32)
        _capi.check(st, "scnerf_mlp_layout_info")
        exp = [lay.fwd_stream, lay.fwd_bias, lay.fwd_bias_f, lay.fwd_bias_v, lay.fwd_bias_rgb, lay.fwd_alpha_w,
               lay.fwd_alpha_b, lay.fwd_total, lay.bwd_stream, lay.bwd_alpha_w, lay.bwd_total,
               lay.save_floats_per_sample, ML.GRAD_FLOATS_PER_SAMPLE]
        for P in (1, 128, 4097):
            if lib.scnerf_mlp_save_floats(pd, P) != lay.save_floats(P) or lib.scnerf_mlp_grad_floats(P) != ML.grad_floats(P):
                raise RuntimeError("workspace size formulas disagree between the kernels and mlp_layout.py")
        if out[:len(exp)].tolist() != exp or int(out[20]) != lay.n_params or int(out[21]) != lay.e_width:
            raise RuntimeError("kernel / mlp_layout.py constants disagree (pd=%d): %s vs %s"
                               % (pd, out[:22].tolist(), exp))


def pack_weights(flat_params: Tensor, kind: str = "fwd", out: Optional[Tensor] = None, pd: int = 3,
                 remap=None) -> Tensor:
    """flat parameter buffer (595 844 floats for pd = 3, 606 596 for pd = 4) -> packed streaming buffer."""
    _f(flat_params, "flat_params")
    lay = ML.layout(pd)
    if flat_params.numel() != lay.n_params:
        raise ValueError("expected %d parameters, got %d" % (lay.n_params, flat_params.numel()))
    idx = _device_index(kind, flat_params.device, pd, remap)
    if out is None:
        out = torch.empty(idx.numel(), dtype=torch.float32, device=flat_params.device)
    st = _capi.load().scnerf_gather_f32(_p(flat_params), _p(idx), _p(out), idx.numel(), _stream())
    _capi.check(st, "scnerf_gather_f32")
    return out


MLP_ARITHMETICS = ("fp32", "resident")
_MLP_ARITHMETIC = [os.environ.get("SCNERF_MLP_ARITHMETIC", "resident")]
if _MLP_ARITHMETIC[0] not in MLP_ARITHMETICS:
    raise ValueError("SCNERF_MLP_ARITHMETIC must be one of %s, not %r" % ("|".join(MLP_ARITHMETICS), _MLP_ARITHMETIC[0]))


def mlp_arithmetic(mode: Optional[str] = None) -> str:
    """How the network runs, training and inference, forward and data gradients: "resident" (the default) -- one launch
    per pass on three fp16 products per product with the activations register-resident as cut fp16 planes
    (csrc/mlp_h3.h): the activation workspace is only written, for the weight-gradient GEMMs; "fp32" -- the fused
    kernels on the exact-fp32 MFMA (csrc/mlp_fwd.hip, mlp_bwd.hip): the numerical yardstick.
    Without an argument: the mode in force.  Environment preset: SCNERF_MLP_ARITHMETIC."""
    if mode is not None:
        if mode not in MLP_ARITHMETICS:
            raise ValueError("mlp_arithmetic is one of " + ", ".join(MLP_ARITHMETICS))
        _MLP_ARITHMETIC[0] = mode
    return _MLP_ARITHMETIC[0]


def pack_for_arithmetic(flat_params: Tensor, train: bool, pd: int = 3, remap=None):
    """What mlp_fwd / coarse_stage_fwd / mlp_bwd take as `planes` in the arithmetic in force: the ResidentWeights
    ("resident") or None ("fp32": the fused fp32-MFMA kernels read the packed fp32 tables only)."""
    if _MLP_ARITHMETIC[0] == "resident":
        return pack_resident(flat_params, pd, remap=remap)
    return None



_PACK_CACHE = [os.environ.get("SCNERF_PACK_CACHE", "1") != "0"]
PACK_CACHE_STATS = {"hits": 0, "packs": 0}


def weight_pack_cache(on: Optional[bool] = None) -> bool:
    """Inference keeps the packed weights of a network between calls (a version-keyed cache):
    render_path renders an image in ~24 chunks of rays with the SAME weights, and re-packing per chunk was 2 x 5 launches
    of nothing.  The key is everything autograd knows about the weights' identity: the flat buffer's address and version
    and every parameter's version (in-place optimizer steps, load_state_dict and FusedAdam.step all bump them).  Writes
    that by-pass version counting (`p.data.mul_(...)`) are invisible to it -- switch the cache off around them
    (`weight_pack_cache(False)`, SCNERF_PACK_CACHE=0) or call `forget_packs(net)`.  Training never uses it."""
    if on is not None:
        _PACK_CACHE[0] = bool(on)
    return _PACK_CACHE[0]


def forget_packs(net) -> None:
    if hasattr(net, "_infer_packs"):
        net._infer_packs = None


def inference_packs(net, flat_params: Tensor, pd: int = 3, remap=None):
    """-> (packed fp32 forward tables, what the arithmetic in force needs besides them) for a forward-only call, packed
    once per weight version (weight_pack_cache)."""
    def build():
        PACK_CACHE_STATS["packs"] += 1
        return (pack_weights(flat_params, "fwd", pd=pd, remap=remap), pack_for_arithmetic(flat_params, False, pd, remap=remap))
    if not _PACK_CACHE[0]:
        return build()
    key = (flat_params.data_ptr(), flat_params._version, tuple(p._version for p in net.parameters()), _MLP_ARITHMETIC[0], pd,
           torch.cuda.current_stream(flat_params.device).cuda_stream if flat_params.is_cuda else 0)
    hit = getattr(net, "_infer_packs", None)
    if hit is not None and hit[0] == key:
        PACK_CACHE_STATS["hits"] += 1
        return hit[1]
    packs = build()
    # (a plain attribute, not a buffer: it must not travel in state_dict() or follow .to())
    object.__setattr__(net, "_infer_packs", (key, packs))
    return packs

_canon_cache = {}


class ResidentWeights:
    """What the resident arithmetic reads besides the packed fp32 tables: the two fp16 fragment streams and the
    scale table (pack_resident)."""
    __slots__ = ("fwd", "bwd", "scales", "pd")

    def __init__(self, fwd, bwd, scales, pd):
        self.fwd, self.bwd, self.scales, self.pd = fwd, bwd, scales, pd


class ChunkMaxima:
    """Per weight-gradient workgroup chunk, the largest |value| of the operands of the eight 256 x 256 GEMMs of one
    network pass: `x` [8, chunks] left by the resident forward, `z` [8, chunks] by the resident data-gradient chain
    (one atomic max per wave and layer); with them the GEMMs run on three fp16 products (csrc/wgrad256_half.h)."""
    __slots__ = ("x", "z", "chunks", "chunk_samples", "scales")

    def __init__(self, P: int, device):
        self.chunks = int(_capi.load().scnerf_wgrad256_chunks(wgrad_chunks(P)))
        self.chunk_samples = int(_capi.load().scnerf_wgrad_chunk_samples(int(P), self.chunks))
        # rows 0 .. 7: the eight 256 x 256 GEMMs; rows 8 .. 11 of z: dZ of the views layer, dZ of layer 0, max(1, |point|),
        # max(1, |direction|) (left by the resident data-gradient kernel for the narrow GEMMs, csrc/wgrad_half_narrow.h)
        both = torch.zeros((2, 12, self.chunks), dtype=torch.float32, device=device)
        self.x, self.z = both[0], both[1]
        self.scales = None        # the scale table of the weights the data-gradient kernel ran with (mlp_bwd_resident)


_h3_tables = {}


def _h3_device_tables(pd: int, device):
    key = (pd, str(device))
    if key not in _h3_tables:
        t = {"jobs": torch.from_numpy(np.ascontiguousarray(ML.h3_scale_jobs(pd))).to(device)}
        for kind in ("fwd", "bwd"):
            idx, meta, _, _ = ML.h3_plan(pd, kind)
            t[kind] = (torch.from_numpy(idx).to(device), torch.from_numpy(meta).to(device), int(meta.shape[0]))
        _h3_tables[key] = t
    return _h3_tables[key]


def pack_resident(flat_params: Tensor, pd: int = 3, out: Optional[ResidentWeights] = None, remap=None) -> ResidentWeights:
    """flat parameter buffer (reference order) -> the fp16 fragment streams of the forward and the data-gradient chain
    and the per-layer scale table of the resident arithmetic (csrc/mlp_h3.h); once per optimizer step.  `remap` as in
    pack_weights."""
    _f(flat_params, "flat_params")
    lay = ML.layout(pd)
    if flat_params.numel() != lay.n_params:
        raise ValueError("expected %d parameters, got %d" % (lay.n_params, flat_params.numel()))
    lib = _capi.load()
    dev = flat_params.device
    if remap is not None:
        key = (remap[0], str(dev))
        if key not in _canon_cache:
            _canon_cache[key] = torch.from_numpy(np.asarray(remap[1], dtype=np.int32)).to(dev)
        idx = _canon_cache[key]
        canon = torch.empty(lay.n_params, dtype=torch.float32, device=dev)
        _capi.check(lib.scnerf_gather_f32(_p(flat_params), _p(idx), _p(canon), idx.numel(), _stream()), "scnerf_gather_f32")
        flat_params = canon
    t = _h3_device_tables(pd, dev)
    if out is None:
        out = ResidentWeights(torch.empty(t["fwd"][2] * 512, dtype=torch.int16, device=dev),
                              torch.empty(t["bwd"][2] * 512, dtype=torch.int16, device=dev),
                              torch.empty(lib.scnerf_h3_scale_floats(), dtype=torch.float32, device=dev), pd)
    st = lib.scnerf_h3_pack(_p(flat_params), _p(t["jobs"]), _p(t["fwd"][0]), _p(t["fwd"][1]), t["fwd"][2],
                            _p(t["bwd"][0]), _p(t["bwd"][1]), t["bwd"][2], _p(out.fwd), _p(out.bwd), _p(out.scales), _stream())
    _capi.check(st, "scnerf_h3_pack")
    return out


def _vd(viewdirs: Tensor):
    """(pointer, row stride) of a [n,3] fp32 view-direction tensor that may be a column slice
    of the packed ray batch (ray_batch[:, 8:11])."""
    if viewdirs.dtype != torch.float32 or not _capi.on_device(viewdirs) or viewdirs.dim() != 2 or viewdirs.shape[1] != 3:
        raise TypeError("viewdirs must be a CUDA fp32 [n,3] tensor")
    if viewdirs.stride(1) != 1:
        raise ValueError("viewdirs rows must be dense")
    return viewdirs.data_ptr(), int(viewdirs.stride(0)) if viewdirs.shape[0] > 1 else 3


_MAC_PER_SAMPLE = {3: 593408, 4: 593408 + 2 * 256 * 21}       # layer 0 and the skip layer are 21 columns wider


def mlp_fwd(pts: Tensor, viewdirs: Tensor, samples_per_ray: int, wpacked: Tensor,
            save: Optional[Tensor] = None, pd: int = 3, planes: Optional[Tensor] = None,
            maxima: Optional["ChunkMaxima"] = None) -> Tensor:
    """pts [P, pd] (pd = 3: x y z; pd = 4: x y z 1/r) -> raw [P, 4] (rgb logits, sigma pre-activation).
    `planes`: the ResidentWeights (pack_resident) -> the resident kernel; None -> the fused fp32-MFMA kernel."""
    _f(pts, "pts"), _f(wpacked, "wpacked")
    vptr, vstride = _vd(viewdirs)
    lay = ML.layout(pd)
    P = pts.numel() // pd
    if wpacked.numel() != lay.fwd_total:
        raise ValueError("wpacked has the wrong size")
    if save is not None:
        _f(save, "save")
        if save.numel() < lay.save_floats(P):
            raise ValueError("activation workspace too small")
    if isinstance(planes, ResidentWeights):
        if planes.pd != pd:
            raise ValueError("resident weights of another network variant")
        return mlp_fwd_resident(pts, viewdirs, samples_per_ray, wpacked, planes, save, maxima)
    raw = torch.empty((P, 4), dtype=torch.float32, device=pts.device)
    tag = "" if pd == 3 else "/pd4"
    if planes is not None:
        raise TypeError("planes must be a ResidentWeights or None")
    with PROFILE.region("mlp_fwd_kernel%s/P=%d/%s" % (tag, P, "train" if save is not None else "infer"),
                        2 * _MAC_PER_SAMPLE[pd] * P):
        st = _capi.load().scnerf_mlp_fwd(pd, _p(pts), vptr, vstride, int(samples_per_ray), _p(wpacked), _p(raw),
                                         _p(save), P, _stream())
    _capi.check(st, "scnerf_mlp_fwd")
    return raw


def mlp_fwd_resident(pts: Tensor, viewdirs: Tensor, samples_per_ray: int, wpacked: Tensor, rw: ResidentWeights,
                     save: Optional[Tensor] = None, maxima: Optional[ChunkMaxima] = None) -> Tensor:
    """mlp_fwd in the resident arithmetic: one launch, three fp16 products per product, activations register-resident
    (csrc/mlp_fwd_h3.hip); save (training) receives the same workspace as mlp_fwd's."""
    _f(pts, "pts"), _f(wpacked, "wpacked")
    vptr, vstride = _vd(viewdirs)
    pd = rw.pd
    lay = ML.layout(pd)
    P = pts.numel() // pd
    if wpacked.numel() != lay.fwd_total:
        raise ValueError("wpacked has the wrong size")
    if save is not None:
        _f(save, "save")
        if save.numel() < lay.save_floats(P):
            raise ValueError("activation workspace too small")
    raw = torch.empty((P, 4), dtype=torch.float32, device=pts.device)
    with PROFILE.region("mlp_fwd_h3_kernel%s/P=%d/%s" % ("" if pd == 3 else "/pd4", P, "train" if save is not None else "infer"),
                        2 * _MAC_PER_SAMPLE[pd] * P):
        mx = maxima if save is not None else None
        st = _capi.load().scnerf_mlp_fwd_h3(pd, _p(pts), vptr, vstride, int(samples_per_ray), _p(wpacked), _p(rw.fwd),
                                            _p(rw.scales), _p(raw), _p(save), P, _p(mx.x) if mx else None,
                                            mx.chunks if mx else 0, mx.chunk_samples if mx else 0, _stream())
    _capi.check(st, "scnerf_mlp_fwd_h3")
    return raw


COARSE_STAGE_SAMPLES = 64        # the fused coarse stage exists for two wave tiles per ray (csrc/mlp_fwd.hip)


def coarse_stage_fwd(rays: Tensor, t_vals: Tensor, t_rand: Optional[Tensor], lindisp: bool, wpacked: Tensor,
                     save: Optional[Tensor], noise: Optional[Tensor], white_bkgd: bool, planes: Optional[Tensor] = None,
                     maxima: Optional["ChunkMaxima"] = None):
    """coarse_sample + mlp_fwd + composite_fwd of the coarse stage as one launch (64 samples per ray):
    -> (z [n,64], pts [n,64,3], raw [n,64,4], rgb [n,3], disp [n], acc [n], weights [n,64], depth [n])."""
    _f(rays, "rays"), _f(t_vals, "t_vals"), _f(wpacked, "wpacked")
    for name, t_ in (("t_rand", t_rand), ("noise", noise), ("save", save)):
        if t_ is not None:
            _f(t_, name)
    n, s = rays.shape[0], t_vals.shape[0]
    if s != COARSE_STAGE_SAMPLES or rays.shape[1] < 11:
        raise ValueError("the fused coarse stage takes 64 samples per ray and an 11-column ray batch")
    lay = ML.layout(3)
    if wpacked.numel() != lay.fwd_total:
        raise ValueError("wpacked has the wrong size")
    if save is not None and save.numel() < lay.save_floats(n * s):
        raise ValueError("activation workspace too small")
    dev = rays.device
    z = torch.empty((n, s), dtype=torch.float32, device=dev)
    pts = torch.empty((n, s, 3), dtype=torch.float32, device=dev)
    raw = torch.empty((n, s, 4), dtype=torch.float32, device=dev)
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((n,), dtype=torch.float32, device=dev)
    acc = torch.empty((n,), dtype=torch.float32, device=dev)
    depth = torch.empty((n,), dtype=torch.float32, device=dev)
    w = torch.empty((n, s), dtype=torch.float32, device=dev)
    P = n * s
    if isinstance(planes, ResidentWeights):
        if planes.pd != 3:
            raise ValueError("the coarse stage samples 3-D points")
        with PROFILE.region("mlp_fwd_h3_kernel<coarse stage>/P=%d/%s" % (P, "train" if save is not None else "infer"),
                            2 * _MAC_PER_SAMPLE[3] * P):
            st = _capi.load().scnerf_coarse_stage_fwd_h3(
                _p(rays), rays.shape[1], _p(t_vals), _p(t_rand), int(bool(lindisp)), _p(wpacked), _p(planes.fwd),
                _p(planes.scales), _p(save), _p(noise), int(bool(white_bkgd)), _p(z), _p(pts), _p(raw), _p(rgb), _p(disp),
                _p(acc), _p(depth), _p(w), n, s, _p(maxima.x) if (maxima and save is not None) else None,
                maxima.chunks if maxima else 0, maxima.chunk_samples if maxima else 0, _stream())
        _capi.check(st, "scnerf_coarse_stage_fwd_h3")
        return z, pts, raw, rgb, disp, acc, w, depth
    if planes is not None:
        raise TypeError("planes must be a ResidentWeights or None")
    with PROFILE.region("mlp_fwd_kernel/P=%d/%s" % (P, "train" if save is not None else "infer"), 2 * _MAC_PER_SAMPLE[3] * P):
        st = _capi.load().scnerf_coarse_stage_fwd(_p(rays), rays.shape[1], _p(t_vals), _p(t_rand), int(bool(lindisp)),
                                                  _p(wpacked), _p(save), _p(noise), int(bool(white_bkgd)), _p(z), _p(pts),
                                                  _p(raw), _p(rgb), _p(disp), _p(acc), _p(depth), _p(w), n, s, _stream())
    _capi.check(st, "scnerf_coarse_stage_fwd")
    return z, pts, raw, rgb, disp, acc, w, depth


FINE_STAGE_IMPORTANCE = (64, 128, 192)       # the fused fine stage exists for 64 + N_importance = 128, 192, 256 samples per ray
# Whether render_rays takes the fine stage as ONE launch (fine_stage_fwd) or as three (fine_sample, mlp_fwd, composite_fwd).
# Both routes are the same device code on the same numbers (bit-identical outputs: tests).  Measured on the MI355X at
# 4096 x (64 + 128) (profiles/r04_fused_fine_stage.txt): one launch 3.14 ms, the three launches back to back 3.07 ms
# (3.45 against 3.20 ms before the samplers' serial sections were rewritten).  The sampler is ~8 us of one wave's work per
# workgroup; as a kernel of its own a CU overlaps sixteen rays of it (43 us for the batch), in front of the resident
# network -- one wave per SIMD, nothing else resident -- every microsecond of it is exposed, eight workgroups in a row per
# CU.  Off by default for that reason; SCNERF_FUSED_FINE_STAGE=1 switches it on.
_FUSED_FINE_STAGE = [os.environ.get("SCNERF_FUSED_FINE_STAGE", "0") not in ("", "0")]


def fused_fine_stage(on: Optional[bool] = None) -> bool:
    if on is not None:
        _FUSED_FINE_STAGE[0] = bool(on)
    return _FUSED_FINE_STAGE[0]


def fine_stage_fwd(rays: Tensor, z_c: Tensor, w_c: Tensor, u: Tensor, wpacked: Tensor, save: Optional[Tensor],
                   noise: Optional[Tensor], white_bkgd: bool, planes: "ResidentWeights", maxima: Optional["ChunkMaxima"] = None,
                   want_inds=False, want_cdf=False, want_weights=False):
    """fine_sample + mlp_fwd + composite_fwd of the fine stage as ONE launch (resident arithmetic; 64 coarse samples and
    N_importance in FINE_STAGE_IMPORTANCE): -> (z_f [n,tot], pts_f [n,tot,3], z_samples [n,sf], z_std [n], inds, cdf,
    raw [n,tot,4], rgb [n,3], disp [n], acc [n], depth [n], weights)."""
    _f(rays, "rays"), _f(z_c, "z_c"), _f(w_c, "w_c"), _f(u, "u"), _f(wpacked, "wpacked")
    for name, t_ in (("noise", noise), ("save", save)):
        if t_ is not None:
            _f(t_, name)
    if not isinstance(planes, ResidentWeights) or planes.pd != 3:
        raise TypeError("the fused fine stage runs on the resident arithmetic (3-D points)")
    n, sc = z_c.shape
    sf = u.shape[-1]
    if sc != COARSE_STAGE_SAMPLES or sf not in FINE_STAGE_IMPORTANCE or rays.shape[1] < 11:
        raise ValueError("the fused fine stage takes 64 coarse samples, N_importance in %s and an 11-column ray batch" % (FINE_STAGE_IMPORTANCE,))
    tot = sc + sf
    stride = sf if u.dim() == 2 else 0
    lay = ML.layout(3)
    if wpacked.numel() != lay.fwd_total:
        raise ValueError("wpacked has the wrong size")
    if save is not None and save.numel() < lay.save_floats(n * tot):
        raise ValueError("activation workspace too small")
    dev = rays.device
    z_f = torch.empty((n, tot), dtype=torch.float32, device=dev)
    pts_f = torch.empty((n, tot, 3), dtype=torch.float32, device=dev)
    z_s = torch.empty((n, sf), dtype=torch.float32, device=dev)
    z_std = torch.empty((n,), dtype=torch.float32, device=dev)
    inds = torch.empty((n, sf), dtype=torch.int64, device=dev) if want_inds else None
    cdf = torch.empty((n, sc - 1), dtype=torch.float32, device=dev) if want_cdf else None
    raw = torch.empty((n, tot, 4), dtype=torch.float32, device=dev)
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((n,), dtype=torch.float32, device=dev)
    acc = torch.empty((n,), dtype=torch.float32, device=dev)
    depth = torch.empty((n,), dtype=torch.float32, device=dev)
    w = torch.empty((n, tot), dtype=torch.float32, device=dev) if want_weights else None
    P = n * tot
    mx = maxima if save is not None else None
    with PROFILE.region("mlp_fwd_h3_kernel<fine stage>/P=%d/%s" % (P, "train" if save is not None else "infer"),
                        2 * _MAC_PER_SAMPLE[3] * P):
        st = _capi.load().scnerf_fine_stage_fwd_h3(
            _p(rays), rays.shape[1], _p(z_c), _p(w_c), _p(u), stride, _p(wpacked), _p(planes.fwd), _p(planes.scales), _p(save),
            _p(noise), int(bool(white_bkgd)), _p(z_f), _p(pts_f), _p(z_s), _p(z_std), _p(inds), _p(cdf), _p(raw), _p(rgb),
            _p(disp), _p(acc), _p(depth), _p(w), n, sc, sf, _p(mx.x) if mx else None, mx.chunks if mx else 0,
            mx.chunk_samples if mx else 0, _stream())
    _capi.check(st, "scnerf_fine_stage_fwd_h3")
    return z_f, pts_f, z_s, z_std, inds, cdf, raw, rgb, disp, acc, depth, w


def save_workspace(P: int, device, pd: int = 3) -> Tensor:
    return torch.empty(ML.layout(pd).save_floats(P), dtype=torch.float32, device=device)


def mlp_bwd(d_raw: Tensor, pts: Tensor, viewdirs: Tensor, samples_per_ray: int, wpacked_bwd: Tensor,
            save: Tensor, pd: int = 3, planes: Optional[Tensor] = None, maxima: Optional["ChunkMaxima"] = None):
    """-> (grads workspace, d_pts [P,pd], d_views [P,3]).  `planes`: as mlp_fwd."""
    if isinstance(planes, ResidentWeights):
        if planes.pd != pd:
            raise ValueError("resident weights of another network variant")
        return mlp_bwd_resident(d_raw, pts, viewdirs, samples_per_ray, wpacked_bwd, planes, save, maxima)
    _f(d_raw, "d_raw"), _f(pts, "pts"), _f(wpacked_bwd, "wpacked_bwd"), _f(save, "save")
    vptr, vstride = _vd(viewdirs)
    lay = ML.layout(pd)
    P = pts.numel() // pd
    if wpacked_bwd.numel() != lay.bwd_total:
        raise ValueError("wpacked_bwd has the wrong size")
    dev = pts.device
    grads = torch.empty(ML.grad_floats(P), dtype=torch.float32, device=dev)
    d_pts = torch.empty((P, pd), dtype=torch.float32, device=dev)
    d_views = torch.empty((P, 3), dtype=torch.float32, device=dev)
    if planes is not None:
        raise TypeError("planes must be a ResidentWeights or None")
    with PROFILE.region("mlp_bwd_kernel%s/P=%d" % ("" if pd == 3 else "/pd4", P), 2 * _MAC_PER_SAMPLE[pd] * P):
        st = _capi.load().scnerf_mlp_bwd(pd, _p(d_raw), _p(pts), vptr, vstride, int(samples_per_ray),
                                         _p(wpacked_bwd), _p(save), _p(grads), _p(d_pts), _p(d_views), P, _stream())
    _capi.check(st, "scnerf_mlp_bwd")
    return grads, d_pts, d_views


def mlp_bwd_resident(d_raw: Tensor, pts: Tensor, viewdirs: Tensor, samples_per_ray: int, wpacked_bwd: Tensor,
                     rw: ResidentWeights, save: Tensor, maxima: Optional[ChunkMaxima] = None):
    """mlp_bwd in the resident arithmetic (csrc/mlp_bwd_h3.hip): one launch -> (grads workspace, d_pts, d_views)."""
    _f(d_raw, "d_raw"), _f(pts, "pts"), _f(wpacked_bwd, "wpacked_bwd"), _f(save, "save")
    vptr, vstride = _vd(viewdirs)
    pd = rw.pd
    lay = ML.layout(pd)
    P = pts.numel() // pd
    if wpacked_bwd.numel() != lay.bwd_total:
        raise ValueError("wpacked_bwd has the wrong size")
    dev = pts.device
    grads = torch.empty(ML.grad_floats(P), dtype=torch.float32, device=dev)
    d_pts = torch.empty((P, pd), dtype=torch.float32, device=dev)
    d_views = torch.empty((P, 3), dtype=torch.float32, device=dev)
    with PROFILE.region("mlp_bwd_h3_kernel%s/P=%d" % ("" if pd == 3 else "/pd4", P), 2 * _MAC_PER_SAMPLE[pd] * P):
        if maxima is not None:
            maxima.scales = rw.scales
        st = _capi.load().scnerf_mlp_bwd_h3(pd, _p(d_raw), _p(pts), vptr, vstride, int(samples_per_ray), _p(wpacked_bwd),
                                            _p(rw.bwd), _p(rw.scales), _p(save), _p(grads), _p(d_pts), _p(d_views), P,
                                            _p(maxima.z) if maxima else None, maxima.chunks if maxima else 0,
                                            maxima.chunk_samples if maxima else 0, _stream())
    _capi.check(st, "scnerf_mlp_bwd_h3")
    return grads, d_pts, d_views


def wgrad_chunks(P: int) -> int:
    """workgroups the sample axis is split into (one per CU at full size)."""
    return int(max(1, min(256, P // 256)))


_wgrad_ws = {}


def wgrad_arithmetic(mode: Optional[str] = None) -> str:
    """The 256 x 256 weight-gradient GEMMs and the narrow ones with a tile-native dZ: "half" (default) -- three fp16
    products per product with one power-of-two scale per operand and workgroup chunk, where the resident kernels left
    the chunk maxima (csrc/wgrad256_half.h, wgrad_half_narrow.h); "fp32", and wherever no maxima were left: the
    exact-fp32 MFMA.  Without an argument: the mode in force."""
    code = {None: -1, "fp32": 0, "half": 1}[mode]
    return ("fp32", "half")[_capi.load().scnerf_wgrad_arithmetic(code)]


def nerf_wgrad(save: Tensor, grads: Tensor, d_raw: Tensor, P: int, flat_grad: Optional[Tensor] = None,
               pd: int = 3, accumulate: bool = False, maxima: Optional[ChunkMaxima] = None) -> Tensor:
    """All parameter gradients of one network -> flat buffer (mlp_layout.Layout parameter order);
    `accumulate`: add to `flat_grad` instead of overwriting it."""
    lib = _capi.load()
    chunks = wgrad_chunks(P)
    key = (chunks, str(save.device))
    if key not in _wgrad_ws:
        _wgrad_ws[key] = torch.empty(lib.scnerf_nerf_wgrad_workspace_floats(chunks), dtype=torch.float32,
                                     device=save.device)
    if flat_grad is None:
        flat_grad = torch.empty(ML.layout(pd).n_params, dtype=torch.float32, device=save.device)
    elif flat_grad.numel() != ML.layout(pd).n_params or flat_grad.dtype != torch.float32 or not flat_grad.is_contiguous():
        raise ValueError("flat_grad must be a contiguous fp32 buffer of %d elements" % ML.layout(pd).n_params)
    tag = "" if pd == 3 else "/pd4"
    # the maxima count only when BOTH resident kernels of this pass filled them (the data-gradient kernel leaves its mark
    # in `scales`): a half-filled table would scale one operand by 2^126
    if maxima is not None and maxima.scales is None:
        maxima = None
    with PROFILE.region("wgrad(12 GEMMs + reduces)%s/P=%d" % (tag, P), 2 * _MAC_PER_SAMPLE[pd] * P, group=True):
        ev = (None, None)
        if PROFILE.enabled:
            # the dominant launch inside this C call -- the eight 256 x 256 GEMMs -- between two events the call records
            mode = wgrad_arithmetic()
            if mode == "half" and maxima is None:
                mode = "fp32"
            inner = PROFILE.raw_pair("wgrad256_kernel<8 GEMMs, %s>%s/P=%d" % (mode, tag, P), 8 * 2 * 256 * 256 * P)
            ev = (inner[0].cuda_event, inner[1].cuda_event)
        if maxima is not None and maxima.chunks != lib.scnerf_wgrad256_chunks(chunks):
            raise ValueError("chunk maxima of another chunking")
        st = lib.scnerf_nerf_wgrad_h3(pd, _p(save), _p(grads), _p(d_raw), P, chunks, _p(_wgrad_ws[key]),
                                      _p(flat_grad), int(bool(accumulate)), _p(maxima.x) if maxima else None,
                                      _p(maxima.z) if maxima else None, _p(maxima.scales) if maxima else None, ev[0], ev[1],
                                      _stream())
    _capi.check(st, "scnerf_nerf_wgrad")
    return flat_grad


def composite_fwd(raw: Tensor, z: Tensor, rays: Tensor, noise: Optional[Tensor], white_bkgd: bool,
                  want_weights=True):
    _f(raw, "raw"), _f(z, "z"), _f(rays, "rays")
    if noise is not None:
        _f(noise, "noise")
    n, s = z.shape
    dev = z.device
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((n,), dtype=torch.float32, device=dev)
    acc = torch.empty((n,), dtype=torch.float32, device=dev)
    depth = torch.empty((n,), dtype=torch.float32, device=dev)
    w = torch.empty((n, s), dtype=torch.float32, device=dev) if want_weights else None
    st = _capi.load().scnerf_composite_fwd(_p(raw), _p(z), _p(rays), rays.shape[1], _p(noise),
                                           int(bool(white_bkgd)), _p(rgb), _p(disp), _p(acc), _p(depth),
                                           _p(w), n, s, _stream())
    _capi.check(st, "scnerf_composite_fwd")
    return rgb, disp, acc, w, depth


def composite_bwd(raw, z, rays, noise, white_bkgd, g_rgb, g_disp, g_acc, g_depth, g_raw_in, want_d_rays_d=True):
    n, s = z.shape
    dev = z.device
    for name, t_ in (("g_rgb", g_rgb), ("g_disp", g_disp), ("g_acc", g_acc), ("g_depth", g_depth),
                     ("g_raw_in", g_raw_in)):
        if t_ is not None:
            _f(t_, name)
    d_raw = torch.empty((n, s, 4), dtype=torch.float32, device=dev)
    d_rd = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_d_rays_d else None
    st = _capi.load().scnerf_composite_bwd(_p(raw), _p(z), _p(rays), rays.shape[1], _p(noise),
                                           int(bool(white_bkgd)), _p(g_rgb), _p(g_disp), _p(g_acc),
                                           _p(g_depth), _p(g_raw_in), _p(d_raw), _p(d_rd), n, s, _stream())
    _capi.check(st, "scnerf_composite_bwd")
    return d_raw, d_rd


def ray_reduce(d_pts, d_views, z, extra_d, d_rays, accumulate: bool):
    n, s = z.shape
    st = _capi.load().scnerf_ray_reduce(_p(d_pts), _p(d_views), _p(z), _p(extra_d), _p(d_rays),
                                        d_rays.shape[1], int(bool(accumulate)), n, s, _stream())
    _capi.check(st, "scnerf_ray_reduce")
    return d_rays


# ------------------------------------------------------------------------------ camera
def camera_matrices_fwd(intr_init, intr_noise, intr_scale, multiplicative, extr_init, extr_noise, extr_scale):
    """-> K [4,4], E [C,4,4] (scnerf_camera_matrices_fwd)"""
    for name, t_ in (("intr_init", intr_init), ("intr_noise", intr_noise), ("extr_init", extr_init), ("extr_noise", extr_noise)):
        _f(t_, name)
    C = extr_init.shape[0]
    if intr_init.numel() != 4 or intr_noise.numel() != 4 or extr_init.shape != (C, 9) or extr_noise.shape != (C, 9):
        raise ValueError("camera parameters: intrinsics [4], extrinsics [C, 9]")
    K = torch.empty((4, 4), dtype=torch.float32, device=intr_init.device)
    E = torch.empty((C, 4, 4), dtype=torch.float32, device=intr_init.device)
    st = _capi.load().scnerf_camera_matrices_fwd(_p(intr_init), _p(intr_noise), float(intr_scale), int(bool(multiplicative)),
                                                 _p(extr_init), _p(extr_noise), float(extr_scale), C, _p(K), _p(E), _stream())
    _capi.check(st, "scnerf_camera_matrices_fwd")
    return K, E


def camera_matrices_bwd(intr_init, intr_noise, intr_scale, multiplicative, extr_init, extr_noise, extr_scale, g_K, g_E,
                        want_intr=True, want_extr=True):
    """-> d intr_noise [4], d extr_noise [C,9] (None where not wanted); g_K / g_E may be None (= zero)"""
    C = extr_init.shape[0]
    for name, t_ in (("g_K", g_K), ("g_E", g_E)):
        if t_ is not None:
            _f(t_, name)
    if not (want_intr or want_extr):
        return None, None
    dev = intr_init.device
    d_in = torch.empty(4, dtype=torch.float32, device=dev) if want_intr else None
    d_ex = torch.empty((C, 9), dtype=torch.float32, device=dev) if want_extr else None
    st = _capi.load().scnerf_camera_matrices_bwd(_p(intr_init), float(intr_scale), int(bool(multiplicative)), _p(extr_init),
                                                 _p(extr_noise), float(extr_scale), C, _p(g_K), _p(g_E), _p(d_in), _p(d_ex),
                                                 _stream())
    _capi.check(st, "scnerf_camera_matrices_bwd")
    return d_in, d_ex


def _cam_common(cam: dict):
    """(ctypes argument tuple shared by camera fwd / bwd) from a dict of tensors / scalars."""
    import ctypes
    F = ctypes.c_float
    g_o, g_d = cam.get("grid_o"), cam.get("grid_d")
    grid = g_o if g_o is not None else g_d
    gh, gw = (int(grid.shape[0]), int(grid.shape[1])) if grid is not None else (0, 0)
    ext = cam.get("extrinsic")
    n_ext = 0 if ext is None else (1 if ext.dim() == 2 else int(ext.shape[0]))
    idx = cam.get("cam_idx")
    return (_p(cam.get("kps")), _p(idx), int(cam.get("single_idx", 0)), _p(ext), n_ext,
            _p(cam["intr_init"]), _p(cam["intr_noise"]), F(float(cam["intr_scale"])), int(bool(cam["multiplicative"])),
            _p(cam["extr_init"]), _p(cam["extr_noise"]), F(float(cam["extr_scale"])), int(cam["extr_init"].shape[0]),
            _p(g_o), F(float(cam.get("scale_o", 0.0))), _p(g_d), F(float(cam.get("scale_d", 0.0))), gh, gw,
            int(cam["H"]), int(cam["W"]))


def camera_rays_fwd(cam: dict, n: int):
    dev = cam["intr_init"].device
    ro = torch.empty((n, 3), dtype=torch.float32, device=dev)
    rd = torch.empty((n, 3), dtype=torch.float32, device=dev)
    st = _capi.load().scnerf_camera_rays_fwd(*_cam_common(cam), _p(ro), _p(rd), n, _stream())
    _capi.check(st, "scnerf_camera_rays_fwd")
    return ro, rd


def camera_rays_bwd(cam: dict, n: int, g_o: Optional[Tensor], g_d: Optional[Tensor]):
    lib = _capi.load()
    dev = cam["intr_init"].device
    C = int(cam["extr_init"].shape[0])
    ext = cam.get("extrinsic")
    n_ext = 0 if ext is None else (1 if ext.dim() == 2 else int(ext.shape[0]))
    d_in = torch.empty(4, dtype=torch.float32, device=dev)
    d_ex = torch.empty((C, 9), dtype=torch.float32, device=dev) if ext is None else None
    d_go = torch.empty_like(cam["grid_o"]) if cam.get("grid_o") is not None else None
    d_gd = torch.empty_like(cam["grid_d"]) if cam.get("grid_d") is not None else None
    d_E = torch.empty((n_ext, 4, 4), dtype=torch.float32, device=dev) if n_ext else None
    ws = torch.empty(lib.scnerf_camera_bwd_workspace_floats(max(C, n_ext)), dtype=torch.float32, device=dev)
    st = lib.scnerf_camera_rays_bwd(*_cam_common(cam), _p(g_o), _p(g_d), _p(d_in), _p(d_ex), _p(d_go), _p(d_gd),
                                    _p(d_E), _p(ws), n, _stream())
    _capi.check(st, "scnerf_camera_rays_bwd")
    return d_in, d_ex, d_go, d_gd, d_E


def pinhole_rays(kps: Optional[Tensor], c2w: Tensor, focal: float, H: int, W: int):
    import ctypes
    _f(c2w, "c2w")
    n = H * W if kps is None else kps.shape[0]
    ro = torch.empty((n, 3), dtype=torch.float32, device=c2w.device)
    rd = torch.empty((n, 3), dtype=torch.float32, device=c2w.device)
    st = _capi.load().scnerf_pinhole_rays(_p(kps), 0 if kps is None else int(kps.shape[1]), _p(c2w),
                                          ctypes.c_float(float(focal)), H, W, _p(ro), _p(rd), n, _stream())
    _capi.check(st, "scnerf_pinhole_rays")
    return ro, rd


def ndc_fwd(H, W, f2: Tensor, near: float, o: Tensor, d: Tensor):
    import ctypes
    no, nd = torch.empty_like(o), torch.empty_like(d)
    st = _capi.load().scnerf_ndc_fwd(H, W, _p(f2), ctypes.c_float(float(near)), _p(o), _p(d), _p(no), _p(nd),
                                     o.shape[0], _stream())
    _capi.check(st, "scnerf_ndc_fwd")
    return no, nd


def ndc_bwd(H, W, f2, near, o, d, g_no, g_nd):
    import ctypes
    g_o, g_d = torch.empty_like(o), torch.empty_like(d)
    g_f = torch.empty(2, dtype=torch.float32, device=o.device)
    st = _capi.load().scnerf_ndc_bwd(H, W, _p(f2), ctypes.c_float(float(near)), _p(o), _p(d), _p(g_no), _p(g_nd),
                                     _p(g_o), _p(g_d), _p(g_f), o.shape[0], _stream())
    _capi.check(st, "scnerf_ndc_bwd")
    return g_o, g_d, g_f


def upsample_grid_fwd(grid: Tensor, scale: float, H: int, W: int):
    import ctypes
    _f(grid, "grid")
    out = torch.empty((H * W, 3), dtype=torch.float32, device=grid.device)
    st = _capi.load().scnerf_upsample_grid_fwd(_p(grid), ctypes.c_float(float(scale)), grid.shape[0], grid.shape[1],
                                               H, W, _p(out), _stream())
    _capi.check(st, "scnerf_upsample_grid_fwd")
    return out


def upsample_grid_bwd(g_out: Tensor, scale: float, gh: int, gw: int, H: int, W: int):
    import ctypes
    d = torch.empty((gh, gw, 3), dtype=torch.float32, device=g_out.device)
    st = _capi.load().scnerf_upsample_grid_bwd(_p(g_out), ctypes.c_float(float(scale)), gh, gw, H, W, _p(d), _stream())
    _capi.check(st, "scnerf_upsample_grid_bwd")
    return d


def prd_loss_fwd(kps0, kps1, r0o, r0d, r1o, r1d, K, E2, eps: float, threshold: float, negate_fx: bool,
                 eval_mode: bool):
    """-> (loss [] , n_match [], sums [6]) device tensors; no host sync."""
    import ctypes
    for name, t in (("kps0", kps0), ("kps1", kps1), ("rays0_o", r0o), ("rays0_d", r0d), ("rays1_o", r1o),
                    ("rays1_d", r1d), ("K", K), ("E2", E2)):
        _f(t, name)
    m = kps0.shape[0]
    out = torch.empty(8, dtype=torch.float32, device=kps0.device)
    st = _capi.load().scnerf_prd_loss_fwd(_p(kps0), _p(kps1), _p(r0o), _p(r0d), _p(r1o), _p(r1d), _p(K), _p(E2),
                                          ctypes.c_float(eps), ctypes.c_float(threshold), int(negate_fx),
                                          int(eval_mode), m, _p(out), out[6:].data_ptr(), out[7:].data_ptr(),
                                          _stream())
    _capi.check(st, "scnerf_prd_loss_fwd")
    return out[6], out[7], out[:6]


def prd_loss_bwd(kps0, kps1, r0o, r0d, r1o, r1d, K, E2, eps: float, threshold: float, negate_fx: bool, sums,
                 g_loss: Tensor):
    """-> (g_rays0_o, g_rays0_d, g_rays1_o, g_rays1_d, g_K [4,4], g_E2 [2,4,4])"""
    import ctypes
    m = kps0.shape[0]
    dev = kps0.device
    g = torch.empty((4, m, 3), dtype=torch.float32, device=dev)
    small = torch.empty(16 + 32 + 36, dtype=torch.float32, device=dev)
    g_loss = _f(g_loss.reshape(1).contiguous(), "g_loss")
    st = _capi.load().scnerf_prd_loss_bwd(_p(kps0), _p(kps1), _p(r0o), _p(r0d), _p(r1o), _p(r1d), _p(K), _p(E2),
                                          ctypes.c_float(eps), ctypes.c_float(threshold), int(negate_fx), m,
                                          _p(sums), _p(g_loss), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                                          g[3].data_ptr(), small.data_ptr(), small[16:].data_ptr(),
                                          small[48:].data_ptr(), _stream())
    _capi.check(st, "scnerf_prd_loss_bwd")
    return g[0], g[1], g[2], g[3], small[:16].view(4, 4), small[16:48].view(2, 4, 4)


def prd_filter(kps0, kps1, r0o, r0d, r1o, r1d, K, E2, eps: float, threshold: float, negate_fx: bool) -> Tensor:
    """-> bool [M]: matches that re-project within `threshold` both ways and lie in front of both cameras."""
    import ctypes
    for name, t in (("kps0", kps0), ("kps1", kps1), ("rays0_o", r0o), ("rays0_d", r0d), ("rays1_o", r1o),
                    ("rays1_d", r1d), ("K", K), ("E2", E2)):
        _f(t, name)
    m = kps0.shape[0]
    keep = torch.empty(m, dtype=torch.uint8, device=kps0.device)
    st = _capi.load().scnerf_prd_filter(_p(kps0), _p(kps1), _p(r0o), _p(r0d), _p(r1o), _p(r1d), _p(K), _p(E2),
                                        ctypes.c_float(eps), ctypes.c_float(threshold), int(negate_fx), m,
                                        _p(keep), _stream())
    _capi.check(st, "scnerf_prd_filter")
    return keep.bool()


def embed_fwd(x: Tensor, freqs: Tensor, include_input: bool) -> Tensor:
    """x [n,d] -> [n, d (include_input + 2 F)] (Embedder.embed)."""
    _f(x, "x"), _f(freqs, "freqs")
    n, d = x.shape
    out = torch.empty((n, d * (int(include_input) + 2 * freqs.numel())), dtype=torch.float32, device=x.device)
    st = _capi.load().scnerf_embed_fwd(_p(x), n, d, _p(freqs), freqs.numel(), int(include_input), _p(out), _stream())
    _capi.check(st, "scnerf_embed_fwd")
    return out


def embed_bwd(x: Tensor, g_out: Tensor, freqs: Tensor, include_input: bool) -> Tensor:
    _f(x, "x"), _f(g_out, "g_out"), _f(freqs, "freqs")
    n, d = x.shape
    g_x = torch.empty_like(x)
    st = _capi.load().scnerf_embed_bwd(_p(x), _p(g_out), n, d, _p(freqs), freqs.numel(), int(include_input), _p(g_x),
                                       _stream())
    _capi.check(st, "scnerf_embed_bwd")
    return g_x


def pack_rays_fwd(H, W, f2: Optional[Tensor], ndc_near: float, o: Tensor, d: Tensor, near: float, far: float, cols: int):
    import ctypes
    _f(o, "rays_o"), _f(d, "rays_d")
    n = o.shape[0]
    out = torch.empty((n, cols), dtype=torch.float32, device=o.device)
    st = _capi.load().scnerf_pack_rays_fwd(int(H), int(W), _p(f2), ctypes.c_float(float(ndc_near)), _p(o), _p(d),
                                           ctypes.c_float(float(near)), ctypes.c_float(float(far)), int(cols), _p(out), n,
                                           _stream())
    _capi.check(st, "scnerf_pack_rays_fwd")
    return out


def pack_rays_bwd(H, W, f2: Optional[Tensor], ndc_near: float, o: Tensor, d: Tensor, cols: int, g: Tensor, want_f2: bool):
    import ctypes
    _f(g, "g_ray_batch")
    g_o, g_d = torch.empty_like(o), torch.empty_like(d)
    g_f = torch.empty(2, dtype=torch.float32, device=o.device) if (want_f2 and f2 is not None) else None
    st = _capi.load().scnerf_pack_rays_bwd(int(H), int(W), _p(f2), ctypes.c_float(float(ndc_near)), _p(o), _p(d), int(cols),
                                           _p(g), _p(g_o), _p(g_d), _p(g_f), o.shape[0], _stream())
    _capi.check(st, "scnerf_pack_rays_bwd")
    return g_o, g_d, g_f
