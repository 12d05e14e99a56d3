"""Ray-parallel data parallelism for one node of MI355X: one process per GPU, every rank holds full
replicas of the networks (and the camera model), renders its own shard of the step's rays, and the
gradients are summed with ONE RCCL all-reduce of a flat fp32 buffer per step (2 x 595 844 network
parameters + the camera's ~11k = 4.8 MB; over xGMI that is latency-bound, tens of microseconds,
against >= 25 ms of MFMA work per 4096-ray step -- SURVEY.md section 8e).

The reference does this differently and worse (nn.DataParallel splits *points* and re-replicates the
module every call, NeRF/create_nerf.py:56-69; NeRF++'s gloo DDP never synchronises the camera
gradients, nerfplusplus/create_nerf.py:64-65); neither pattern is reproduced."""
from __future__ import annotations

from typing import Iterable, List

import torch


class FlatGradAllReduce:
    """One flat fp32 gradient buffer and one all-reduce per step.

    Two ways to get the buffer (it must have ONE owner, or a later zero_grad() re-points the .grad views and
    the collective would run on stale memory):
      * `FlatGradAllReduce(modules, world)` allocates it and attaches every parameter's .grad as a view
        (no optimizer involved: bench.py, tests);
      * `FlatGradAllReduce.for_optimizer(optimizer, world)` borrows the gradient arena of a
        scnerf_amd.optim.FusedAdam / CustomAdamOptimizer -- the optimizer stays the owner, `zero()` is its
        zero_grad().  `optimizer.grad_sync = reducer` makes `optimizer.step()` run the collective first, so
        a training loop written for DistributedDataParallel (zero_grad / backward / step) needs no change.

    `all_reduce(local_rays, total_rays)`: each rank's gradient is that of the MEAN loss over its own rays; the
    global-batch mean weights rank r by n_r / N.  Without counts every rank weighs 1 / world (equal shards)."""

    def __init__(self, modules: Iterable[torch.nn.Module], world_size: int, process_group=None):
        self.params: List[torch.nn.Parameter] = []
        seen = set()
        for m in modules:
            for p in m.parameters():
                if id(p) not in seen and p.requires_grad:
                    seen.add(id(p))
                    self.params.append(p)
        self.world_size = int(world_size)
        self.group = process_group
        self.optimizer = None
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self._own = torch.zeros(n, dtype=torch.float32, device=dev)
        self.attach()

    @classmethod
    def for_optimizer(cls, optimizer, world_size: int, process_group=None):
        self = cls.__new__(cls)
        self.params, self._own = None, None
        self.world_size, self.group, self.optimizer = int(world_size), process_group, optimizer
        optimizer.flat_gradient()                      # builds the arena and attaches the views
        return self

    @property
    def flat(self) -> torch.Tensor:
        return self._own if self.optimizer is None else self.optimizer.flat_gradient()

    def attach(self):
        if self.optimizer is not None:
            return self.optimizer.flat_gradient()
        o, self._views = 0, []
        for p in self.params:
            v = self._own[o:o + p.numel()].view(p.shape)
            p.grad = v
            self._views.append(v)
            o += p.numel()

    def zero(self):
        if self.optimizer is not None:
            return self.optimizer.zero_grad()
        self._own.zero_()
        # autograd may have replaced a .grad (set_to_none, a first accumulation into a fresh tensor): compare by identity
        # with the view this object attached -- 52 pointer comparisons, no tensor call per parameter per step
        for p, v in zip(self.params, self._views):
            if p.grad is not v:
                p.grad = v

    def all_reduce(self, local_rays=None, total_rays=None):
        flat = self.flat
        if self.world_size > 1:
            import torch.distributed as dist
            weight = 1.0 / self.world_size if local_rays is None else float(local_rays) / float(total_rays)
            flat.mul_(weight)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat


def shard_rays(n_total: int, rank: int, world_size: int):
    """Contiguous, near-equal ray shards: rank r renders rays [lo, hi)."""
    base, rem = divmod(n_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def render_path_sharded(render_poses, hwf, chunk, render_kwargs, mode, rank: int, world_size: int,
                        process_group=None, render_fn=None, device=None, **path_kwargs):
    """Multi-GPU `render_path` (reference NeRF/render.py:143-183): every rank renders a contiguous band
    of each image's rays ([lo, hi) of the H*W row-major pixels, `shard_rays`), the bands are exchanged
    with ONE all-gather per image (RCCL; rgb+disp packed as 4 floats per pixel, equal-size padded
    shards), and every rank returns the full (rgbs [n,H,W,3], disps [n,H,W]) numpy arrays -- identical
    to the single-GPU result, since rays are independent and evaluation draws no random numbers.
    `path_kwargs`: camera_model, noisy_extrinsic, gt_intrinsic, gt_extrinsic, i_map, transform_align.
    `render_fn` defaults to scnerf_amd.render.render (injectable for the CPU gloo test)."""
    import numpy as np
    import torch.distributed as dist
    if render_fn is None:
        from .render import render as render_fn
    from .render import _image_kwargs
    H, W, _ = hwf
    n_pix = H * W
    lo, hi = shard_rays(n_pix, rank, world_size)
    width = -(-n_pix // world_size)                      # padded shard length
    rgbs, disps = [], []
    ring = None                                           # pinned staging + copy stream (GPU tensors only)

    def collect(img):
        rgbs.append(img[:, :3].reshape(H, W, 3).copy())
        disps.append(img[:, 3].reshape(H, W).copy())

    for i, _pose in enumerate(render_poses):
        kw = _image_kwargs(i, hwf, chunk, render_kwargs, mode, path_kwargs.get("camera_model"),
                           path_kwargs.get("noisy_extrinsic"), path_kwargs.get("gt_intrinsic"),
                           path_kwargs.get("gt_extrinsic"), path_kwargs.get("i_map"),
                           path_kwargs.get("transform_align"))
        with torch.no_grad():
            rgb, disp, _acc, _ = render_fn(_ray_range=(lo, hi), **kw)
        dev = rgb.device if device is None else device
        mine = torch.zeros((width, 4), dtype=torch.float32, device=dev)
        mine[:hi - lo, :3] = rgb.reshape(-1, 3)
        mine[:hi - lo, 3] = disp.reshape(-1)
        if world_size > 1:
            full = torch.empty((world_size, width, 4), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(full.view(-1), mine.view(-1), group=process_group)
            parts = []
            for r in range(world_size):
                a, b = shard_rays(n_pix, r, world_size)
                parts.append(full[r, :b - a])
            img = torch.cat(parts, 0)
        else:
            img = mine[:n_pix]
        if img.is_cuda:
            # as the single-GPU render_path: image i leaves through a pinned buffer on a side stream while
            # image i+1 renders; the host only waits when it needs the pixels
            from .render import _HostRing
            if ring is None:
                ring = _HostRing(img.device)
            ring.push(img)
            if i > 0:
                collect(ring.get(i - 1)[0])
        else:
            collect(img.numpy())
    if ring is not None and ring.pending:
        collect(ring.get(len(ring.pending) - 1)[0])
    return np.stack(rgbs, 0), np.stack(disps, 0)
