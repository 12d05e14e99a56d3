"""Fused Adam on flat buffers.

`FusedAdam` and `CustomAdamOptimizer` are drop-ins for torch.optim.Adam and the reference's
CustomAdamOptimizer (/root/reference NeRF/create_nerf.py:257-335: Adam whose weight decay touches only
the trailing tensors of the stepped list -- ray-origin / ray-direction / distortion noise once they are
learnable --, their number decided by substring tests on `args.camera_model`, :219-226).  Parameters that share one contiguous buffer (every scnerf_amd.NeRF
does: `flat_parameters()`) form one *segment* with one flat gradient buffer (the `.grad`s are views of
it, so autograd accumulates in place and an all-reduce can run on the same memory), one pair of flat
moment buffers and ONE kernel launch per step instead of ~8 element-wise launches per tensor.

LR schedule: `param_groups[i]['lr']` is read at every step, exactly like torch optimizers, so the
reference's `param_group['lr'] = new_lrate` (run_nerf.py:617-621) keeps working; `decayed_lr` is that
formula."""
from __future__ import annotations

from typing import Iterable, List

import torch

from . import _capi


def decayed_lr(lrate: float, lrate_decay: float, global_step: int) -> float:
    """run_nerf.py:617-621."""
    return lrate * (0.1 ** (global_step / (lrate_decay * 1000)))


class _Segment:
    """A run of parameters that tile one contiguous fp32 range and share a step count: one flat view of the
    parameters, one slice of the optimizer's gradient arena, one pair of moment buffers, one kernel launch per step.
    `decay` = (lo, hi): the element range the weight decay applies to (whole tensors; empty when lo == hi) -- a
    decayed tail that starts in the middle of a network's buffer does NOT split the segment (the boundary need not
    be 16-byte aligned, and a split would also break the one-buffer-per-network gradient accumulation)."""

    def __init__(self, params: List[torch.nn.Parameter], decay, step: int, flat_grad: torch.Tensor):
        self.params = params
        self.decay = decay
        self.step = step
        first = params[0]
        n = sum(p.numel() for p in params)
        base = first.data_ptr()
        off = 0
        for p in params:
            assert p.dtype == torch.float32 and p.is_contiguous() and p.data_ptr() == base + 4 * off
            off += p.numel()
        assert flat_grad.numel() == n
        self.n = n
        self.flat_param = torch.as_strided(first.data, (n,), (1,))      # view over the whole range
        self.flat_grad = flat_grad
        self.exp_avg = torch.zeros_like(self.flat_grad)
        self.exp_avg_sq = torch.zeros_like(self.flat_grad)

    def attach(self, keep_existing=False):
        o = 0
        for p in self.params:
            g = p.grad
            want = self.flat_grad.data_ptr() + 4 * o
            if g is None or g.data_ptr() != want:
                view = self.flat_grad[o:o + p.numel()].view(p.shape)
                if g is not None and keep_existing:
                    view.copy_(g)           # a gradient produced before the optimizer (re)built its buffers
                p.grad = view
            o += p.numel()

    def still_valid(self) -> bool:
        base = self.flat_param.data_ptr()
        o = 0
        for p in self.params:
            if p.data_ptr() != base + 4 * o:
                return False
            o += p.numel()
        return True


def _pad16(t: torch.Tensor) -> bool:
    return t.data_ptr() % 16 == 0


class FusedAdam(torch.optim.Optimizer):
    """Adam (no amsgrad) with one fused HIP launch per contiguous parameter segment.

    Weight decay: everywhere when `weight_decay != 0` (torch.optim.Adam), unless `decay_tail = k` restricts it
    to the LAST k tensors among those that currently take part in the step (requires_grad) -- the reference's
    CustomAdamOptimizer rule, which counts inside `params_with_grad` (create_nerf.py:216-226, :290-303): while
    the camera's ray-noise tensors are frozen the decayed tail is therefore made of whatever tensors come last
    among the active ones.

    All gradients live in ONE arena (`flat_gradient()`): every `.grad` is a view of it, so autograd (and the
    weight-gradient kernels) accumulate in place and a ray-parallel all-reduce is one collective on one buffer
    (scnerf_amd.parallel.FlatGradAllReduce.for_optimizer)."""

    def __init__(self, params: Iterable, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 decay_tail=None):
        params = list(params)
        if params and isinstance(params[0], dict):
            raise NotImplementedError("FusedAdam takes one flat parameter list (as the reference does)")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._decay_tail = decay_tail
        self._segments = None
        self._arena = None
        self.grad_sync = None            # a parallel.FlatGradAllReduce.for_optimizer(self, ...): step() all-reduces first
        self._loaded = {}                # id(parameter) -> (step, exp_avg, exp_avg_sq) restored from a checkpoint
        self._step_as_tensor = True      # torch.optim.Adam keeps `step` as a tensor, the reference's class an int

    # -- segments ---------------------------------------------------------------------------------
    def _per_parameter_state(self):
        """(step, exp_avg, exp_avg_sq) of every parameter that has been stepped or restored, keyed by id."""
        out = dict(self._loaded)
        for s in (self._segments or []):
            if s.step == 0:
                continue
            o = 0
            for p in s.params:
                n = p.numel()
                out[id(p)] = (s.step, s.exp_avg[o:o + n], s.exp_avg_sq[o:o + n])
                o += n
        return out

    def _build_segments(self):
        plist = self.param_groups[0]["params"]
        carried = self._per_parameter_state()
        pending = {id(p): p.grad for p in plist if p.requires_grad and p.grad is not None}
        active = [i for i, p in enumerate(plist) if p.requires_grad]
        wd = self.param_groups[0]["weight_decay"]
        if self._decay_tail is None:
            decayed = set(active) if wd != 0 else set()
        else:
            decayed = set(active[len(active) - self._decay_tail:]) if self._decay_tail > 0 else set()
        runs, cur, cur_key, cur_idx = [], [], None, []
        for i, p in enumerate(plist):
            if not p.requires_grad:
                if cur:
                    runs.append((cur, cur_key, cur_idx))
                    cur, cur_key, cur_idx = [], None, []
                continue
            if not _capi.on_device(p):
                raise RuntimeError("FusedAdam needs GPU parameters (scnerf_amd has no CPU path)")
            contiguous = bool(cur) and p.data_ptr() == cur[-1].data_ptr() + 4 * cur[-1].numel()
            # tensors with different step counts (bias corrections differ) cannot share a launch; neither can a
            # decayed tensor FOLLOWED by an undecayed one (the kernel takes one decayed element range per segment)
            key = (p.device, carried[id(p)][0] if id(p) in carried else 0)
            gap = bool(cur) and (cur_idx[-1] in decayed) and (i not in decayed)
            if cur and contiguous and cur_key == key and not gap:
                cur.append(p)
                cur_idx.append(i)
            else:
                if cur:
                    runs.append((cur, cur_key, cur_idx))
                cur, cur_key, cur_idx = [p], key, [i]
        if cur:
            runs.append((cur, cur_key, cur_idx))
        # one gradient arena; every segment's slice starts on a 16-byte boundary
        sizes = [sum(p.numel() for p in ps) for ps, _, _ in runs]
        starts, total = [], 0
        for n in sizes:
            starts.append(total)
            total += (n + 3) // 4 * 4
        dev = runs[0][0][0].device if runs else torch.device("cpu")
        self._arena = torch.zeros(total, dtype=torch.float32, device=dev)
        segs = []
        placed = set()
        for (ps, key, idxs), st, n in zip(runs, starts, sizes):
            # the decayed tensors of a run are its last ones (see `gap` above)
            lo = sum(p.numel() for p, i in zip(ps, idxs) if i not in decayed)
            seg = _Segment(ps, (lo, n), key[1], self._arena[st:st + n])
            placed.update(id(p) for p in ps)
            o = 0
            for p in ps:                                  # moments follow their parameter through a rebuild
                got = carried.get(id(p))
                if got is not None:
                    seg.exp_avg[o:o + p.numel()].copy_(got[1].reshape(-1))
                    seg.exp_avg_sq[o:o + p.numel()].copy_(got[2].reshape(-1))
                g = pending.get(id(p))
                if g is not None:
                    seg.flat_grad[o:o + p.numel()].copy_(g.reshape(-1))
                o += p.numel()
            seg.attach()
            if not _pad16(seg.flat_param):
                raise RuntimeError("parameter segment is not 16-byte aligned")
            segs.append(seg)
        self._segments = segs
        # state of parameters that sit in no segment right now (frozen): kept for the day they are unfrozen
        self._loaded = {pid: (st_, m.clone(), v.clone()) for pid, (st_, m, v) in carried.items() if pid not in placed}
        self._req = [p.requires_grad for p in plist]

    def segments(self):
        plist = self.param_groups[0]["params"]
        if (self._segments is None or self._req != [p.requires_grad for p in plist]
                or not all(s.still_valid() for s in self._segments)):
            self._build_segments()
        return self._segments

    # -- checkpoints: torch.optim's per-parameter format, so files written by the reference load here ----
    def state_dict(self):
        """Same layout as torch.optim.Adam / the reference's CustomAdamOptimizer (`state[i] = {step,
        exp_avg, exp_avg_sq}` per parameter index that has been stepped, `param_groups`): checkpoints
        are interchangeable with the reference's (run_nerf.py:626-641, create_nerf.py:142-172)."""
        self.state.clear()
        by_id = {id(p): p for p in self.param_groups[0]["params"]}
        for pid, (step, m, v) in self._per_parameter_state().items():
            p = by_id.get(pid)
            if p is None:
                continue
            self.state[p] = {"step": torch.tensor(float(step)) if self._step_as_tensor else int(step),
                             "exp_avg": m.reshape(p.shape).clone(), "exp_avg_sq": v.reshape(p.shape).clone()}
        sd = super().state_dict()
        self.state.clear()
        # a checkpoint boundary: look at the key-point batches of the steps so far -- AFTER the state has been assembled, so
        # one bad batch (which the reference would have refused when it was drawn) does not cost the run its checkpoint.
        # The failure is not swallowed: a warning here, and it stays pending -- the next ray-generation call, render_path
        # or interpreter exit raises the reference's AssertionError (SCNERF_SYNC_KEYPOINT_CHECK=1 is the strict mode that
        # raises at the faulty call itself).
        from .get_rays import KEYPOINT_CHECK
        try:
            KEYPOINT_CHECK.flush()
        except AssertionError as e:
            import warnings
            warnings.warn("scnerf_amd.get_rays: %s (found while writing a checkpoint; the checkpoint is written, the "
                          "assertion is raised by the next ray-generation call)" % e, RuntimeWarning)
            KEYPOINT_CHECK.carry(str(e))
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)          # validates the groups, casts tensors to the parameters' device
        loaded = {}
        for p, st in self.state.items():
            if not len(st):
                continue
            if "max_exp_avg_sq" in st:
                raise NotImplementedError("amsgrad state is not supported")
            loaded[id(p)] = (int(float(st["step"])), st["exp_avg"].detach().float().reshape(-1),
                             st["exp_avg_sq"].detach().float().reshape(-1))
        self.state.clear()
        self._segments = None                        # whatever was stepped before is replaced by the file
        self._loaded = loaded
        self.segments()

    def zero_grad(self, set_to_none: bool = False):
        """Zeroes the gradient arena in place (the .grad views stay attached)."""
        segs = self.segments()
        self._arena.zero_()
        for s in segs:
            s.attach()

    def flat_gradients(self):
        """The per-segment gradient buffers (views of the arena)."""
        return [s.flat_grad for s in self.segments()]

    def flat_gradient(self) -> torch.Tensor:
        """The ONE buffer every .grad is a view of -- what a ray-parallel all-reduce sums."""
        self.segments()
        return self._arena

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        beta1, beta2 = g["betas"]
        lib = _capi.load()
        stream = _capi.current_stream()
        segs = self.segments()
        for s in segs:
            s.attach(keep_existing=True)
        if self.grad_sync is not None:
            self.grad_sync.all_reduce()          # ray-parallel ranks: one collective on the arena, then step
        for s in segs:
            s.step += 1
            lo, hi = s.decay
            st = lib.scnerf_adam_step_range(s.flat_param.data_ptr(), s.flat_grad.data_ptr(), s.exp_avg.data_ptr(),
                                            s.exp_avg_sq.data_ptr(), s.n, float(g["lr"]), float(beta1), float(beta2),
                                            float(g["eps"]), float(g["weight_decay"]), lo, hi, s.step, stream)
            _capi.check(st, "scnerf_adam_step_range")
        # the kernel wrote the parameters through raw pointers: tell autograd (what an in-place tensor op does), so that a
        # graph holding an old value refuses a late backward and memos keyed on `_version` (CameraModel._matrices) see the step
        torch.autograd.graph.increment_version([p for p in g["params"] if p.grad is not None])
        return loss


class CustomAdamOptimizer(FusedAdam):
    """Same constructor as the reference (create_nerf.py:257-268); the trailing ray_o / ray_d /
    distortion tensors (by substring of args.camera_model, :219-226) are the only decayed ones."""

    def __init__(self, params, lr, args, H, W, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by SCNeRF and not implemented")
        params = list(params)
        tail = 0
        if args.camera_model != "none":
            tail = ("rayo" in args.camera_model) + ("rayd" in args.camera_model) + ("dist" in args.camera_model)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                         decay_tail=tail if weight_decay != 0 else 0)
        self.args, self.H, self.W = args, H, W
        self._step_as_tensor = False
