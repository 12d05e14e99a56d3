"""Fused Adam on flat buffers.

`FusedAdam` and `CustomAdamOptimizer` are drop-ins for torch.optim.Adam and the reference's
CustomAdamOptimizer (/root/reference NeRF/create_nerf.py:257-335: Adam whose weight decay touches only
the trailing ray-origin / ray-direction / distortion tensors, decided by substring tests on
`args.camera_model`, :219-226).  Parameters that share one contiguous buffer (every scnerf_amd.NeRF
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
    def __init__(self, params: List[torch.nn.Parameter], decay: bool):
        self.params = params
        self.decay = decay
        self.step = 0
        first = params[0]
        n = sum(p.numel() for p in params)
        # the parameters must tile one contiguous fp32 range
        base = first.data_ptr()
        off = 0
        for p in params:
            assert p.dtype == torch.float32 and p.is_contiguous() and p.data_ptr() == base + 4 * off
            off += p.numel()
        self.n = n
        self.flat_param = torch.as_strided(first.data, (n,), (1,))      # view over the whole range
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=first.device)
        self.exp_avg = torch.zeros_like(self.flat_grad)
        self.exp_avg_sq = torch.zeros_like(self.flat_grad)
        self.attach()

    def attach(self):
        o = 0
        for p in self.params:
            g = p.grad
            want = self.flat_grad.data_ptr() + 4 * o
            if g is None or g.data_ptr() != want:
                view = self.flat_grad[o:o + p.numel()].view(p.shape)
                if g is not None:
                    view.copy_(g)           # a gradient produced before the optimizer existed
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

    `decay_from`: index into the parameter list from which weight decay applies (None = nowhere when
    weight_decay == 0, everywhere otherwise ... use CustomAdamOptimizer for the reference's rule)."""

    def __init__(self, params: Iterable, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 decay_from=None):
        params = list(params)
        if params and isinstance(params[0], dict):
            raise NotImplementedError("FusedAdam takes one flat parameter list (as the reference does)")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._decay_from = (0 if weight_decay != 0 else len(params)) if decay_from is None else decay_from
        self._segments = None
        self._loaded_steps = {}          # parameter index -> step count restored from a checkpoint
        self._step_as_tensor = True      # torch.optim.Adam keeps `step` as a tensor, the reference's class an int

    # -- segments ---------------------------------------------------------------------------------
    def _build_segments(self):
        plist = self.param_groups[0]["params"]
        segs, cur, cur_key = [], [], None
        for i, p in enumerate(plist):
            if not p.requires_grad:
                if cur:
                    segs.append(_Segment(cur, cur_key[1]))
                    cur, cur_key = [], None
                continue
            if not _capi.on_device(p):
                raise RuntimeError("FusedAdam needs GPU parameters (scnerf_amd has no CPU path)")
            decay = i >= self._decay_from
            contiguous = bool(cur) and p.data_ptr() == cur[-1].data_ptr() + 4 * cur[-1].numel()
            # tensors restored with different step counts (bias corrections differ) cannot share a launch
            key = (p.device, decay, self._loaded_steps.get(i, 0))
            if cur and contiguous and cur_key == key:
                cur.append(p)
            else:
                if cur:
                    segs.append(_Segment(cur, cur_key[1]))
                cur, cur_key = [p], key
        if cur:
            segs.append(_Segment(cur, cur_key[1]))
        for s in segs:
            if not _pad16(s.flat_param):
                raise RuntimeError("parameter segment is not 16-byte aligned")
        self._segments = segs
        self._req = [p.requires_grad for p in plist]

    def segments(self):
        plist = self.param_groups[0]["params"]
        if (self._segments is None or self._req != [p.requires_grad for p in plist]
                or not all(s.still_valid() for s in self._segments)):
            old = {id(s.params[0]): s for s in (self._segments or [])}
            self._build_segments()
            for s in self._segments:        # keep the moments / step of segments that did not change
                o = old.get(id(s.params[0]))
                if o is not None and o.n == s.n:
                    s.exp_avg, s.exp_avg_sq, s.step = o.exp_avg, o.exp_avg_sq, o.step
        return self._segments

    # -- checkpoints: torch.optim's per-parameter format, so files written by the reference load here ----
    def state_dict(self):
        """Same layout as torch.optim.Adam / the reference's CustomAdamOptimizer (`state[i] = {step,
        exp_avg, exp_avg_sq}` per parameter index that has been stepped, `param_groups`): checkpoints
        are interchangeable with the reference's (run_nerf.py:626-641, create_nerf.py:142-172)."""
        self.state.clear()
        for s in (self._segments or []):
            if s.step == 0:
                continue
            o = 0
            for p in s.params:
                n = p.numel()
                self.state[p] = {
                    "step": torch.tensor(float(s.step)) if self._step_as_tensor else int(s.step),
                    "exp_avg": s.exp_avg[o:o + n].view(p.shape).clone(),
                    "exp_avg_sq": s.exp_avg_sq[o:o + n].view(p.shape).clone()}
                o += n
        sd = super().state_dict()
        self.state.clear()
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)          # validates the groups, casts tensors to the parameters' device
        plist = self.param_groups[0]["params"]
        index = {id(p): i for i, p in enumerate(plist)}
        loaded = {index[id(p)]: st for p, st in self.state.items() if id(p) in index and len(st)}
        self._loaded_steps = {i: int(float(st["step"])) for i, st in loaded.items()}
        self._segments = None
        for s in self.segments():
            o = 0
            for p in s.params:
                st = loaded.get(index[id(p)])
                n = p.numel()
                if st is not None:
                    if "max_exp_avg_sq" in st:
                        raise NotImplementedError("amsgrad state is not supported")
                    s.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
                    s.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                    s.step = int(float(st["step"]))
                o += n
        self.state.clear()

    def zero_grad(self, set_to_none: bool = False):
        """Zeroes the flat gradient buffers in place (the .grad views stay attached)."""
        for s in self.segments():
            s.flat_grad.zero_()
            s.attach()

    def flat_gradients(self):
        """The flat gradient buffers (one per segment) -- what a ray-parallel all-reduce sums."""
        return [s.flat_grad for s in self.segments()]

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
        for s in self.segments():
            s.attach()
            s.step += 1
            wd = g["weight_decay"] if s.decay else 0.0
            st = lib.scnerf_adam_step(s.flat_param.data_ptr(), s.flat_grad.data_ptr(), s.exp_avg.data_ptr(),
                                      s.exp_avg_sq.data_ptr(), s.n, float(g["lr"]), float(beta1), float(beta2),
                                      float(g["eps"]), float(wd), s.step, stream)
            _capi.check(st, "scnerf_adam_step")
        return loss


class CustomAdamOptimizer(FusedAdam):
    """Same constructor as the reference (create_nerf.py:257-268); the trailing ray_o / ray_d /
    distortion tensors (by substring of args.camera_model, :219-226) are the only decayed ones."""

    def __init__(self, params, lr, args, H, W, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by SCNeRF and not implemented")
        params = list(params)
        decay_from = len(params)
        if args.camera_model != "none":
            decay_from -= "rayo" in args.camera_model
            decay_from -= "rayd" in args.camera_model
            decay_from -= "dist" in args.camera_model
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decay_from=decay_from)
        self.args, self.H, self.W = args, H, W
        self._step_as_tensor = False
