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
    """Owns one flat gradient buffer; every parameter's .grad is a view of it, so autograd
    accumulates straight into the buffer the collective runs on."""

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
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.attach()

    def attach(self):
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view(p.shape)
            o += p.numel()

    def zero(self):
        self.flat.zero_()
        # autograd may have replaced a .grad (e.g. after set_to_none): re-point cheaply
        o = 0
        for p in self.params:
            g = p.grad
            if g is None or g.data_ptr() != self.flat.data_ptr() + 4 * o:
                p.grad = self.flat[o:o + p.numel()].view(p.shape)
            o += p.numel()

    def all_reduce(self):
        """sum over ranks, then / world_size (== the gradient of the mean loss over the global batch
        when every rank renders the same number of rays)."""
        if self.world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.mul_(1.0 / self.world_size)
        return self.flat


def shard_rays(n_total: int, rank: int, world_size: int):
    """Contiguous, near-equal ray shards: rank r renders rays [lo, hi)."""
    base, rem = divmod(n_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
