"""Mirror of the reference's `run_nerf_helpers` module (/root/reference NeRF/run_nerf_helpers.py)
for the names the render path and `run_nerf.py` import: `NeRF`, `get_embedder`, `Embedder`,
`DenseLayer`, `img2mse`, `mse2psnr`, `fix_seeds`.

`NeRF` keeps the reference's constructor, parameter names and initialisation, so checkpoints
interoperate, but it does not evaluate layers with torch ops: the render path recognises it and
runs the fused HIP kernels on its *flat* parameter storage (all 24 tensors are views of one
buffer, which is also what the RCCL gradient all-reduce works on).

Unlike the reference this module does NOT switch on torch.autograd.set_detect_anomaly at import
(reference :7 -- a debugging aid that costs ~19 % of every backward)."""
from __future__ import annotations

import random
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _capi
from . import mlp_layout as ML

img2mse = lambda x, y: torch.mean((x - y) ** 2)                                   # reference :10
mse2psnr = lambda x: -10. * torch.log(x) / torch.log(torch.tensor([10.], device=x.device))  # :11


class DenseLayer(nn.Linear):
    """nn.Linear with xavier_uniform(gain of `activation`) weights and zero bias (reference :13-21)."""

    def __init__(self, in_dim: int, out_dim: int, activation: str = "relu", *args, **kwargs) -> None:
        self.activation = activation
        super().__init__(in_dim, out_dim, *args, **kwargs)

    def reset_parameters(self) -> None:
        torch.nn.init.xavier_uniform_(self.weight, gain=torch.nn.init.calculate_gain(self.activation))
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)


class Embedder:
    """Positional-encoding *descriptor* (reference :24-55).  The fused kernels compute the encoding
    in registers; this object only carries (multires, out_dim) so `create_nerf` / `run_network`
    can recognise the standard configuration."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d = kwargs["input_dims"]
        self.num_freqs = kwargs["num_freqs"]
        self.out_dim = (d if kwargs["include_input"] else 0) + d * 2 * self.num_freqs

    def embed(self, inputs):
        raise NotImplementedError(
            "scnerf_amd evaluates the positional encoding inside the fused HIP network kernel; "
            "a stand-alone embedding op is not part of the hot path (use render_rays / run_network)")

    __call__ = embed


def get_embedder(multires, i=0):
    """(embed_fn, out_dim) like the reference (:57-72); embed_fn is an `Embedder` descriptor
    (nn.Identity for i == -1)."""
    if i == -1:
        return nn.Identity(), 3
    eo = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                  log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return eo, eo.out_dim


class NeRF(nn.Module):
    """Same constructor / parameter names as the reference (:76-103)."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views = input_ch, input_ch_views
        self.skips = skips
        self.use_viewdirs = use_viewdirs
        self.pts_linears = nn.ModuleList(
            [DenseLayer(input_ch, W, activation="relu")]
            + [DenseLayer(W, W, activation="relu") if i not in self.skips
               else DenseLayer(W + input_ch, W, activation="relu") for i in range(D - 1)])
        self.views_linears = nn.ModuleList([DenseLayer(input_ch_views + W, W // 2, activation="relu")])
        if use_viewdirs:
            self.feature_linear = DenseLayer(W, W, activation="linear")
            self.alpha_linear = DenseLayer(W, 1, activation="linear")
            self.rgb_linear = DenseLayer(W // 2, 3, activation="linear")
        else:
            self.output_linear = DenseLayer(W, output_ch, activation="linear")
        self._flat: Optional[torch.Tensor] = None

    # ---- fused-kernel support -----------------------------------------------------------
    def is_standard(self) -> bool:
        """The configuration the fused gfx950 kernels are specialised for (every SCNeRF script)."""
        return (self.D == 8 and self.W == 256 and list(self.skips) == [4] and self.use_viewdirs
                and self.input_ch == ML.IN_PTS and self.input_ch_views == ML.IN_VIEWS)

    def require_standard(self):
        if not self.is_standard():
            raise NotImplementedError(
                "the fused MI355X kernels cover the SCNeRF network (D=8, W=256, skips=[4], "
                "use_viewdirs, multires 10/4); got D=%s W=%s skips=%s use_viewdirs=%s in=%s/%s"
                % (self.D, self.W, self.skips, self.use_viewdirs, self.input_ch, self.input_ch_views))

    def ordered_parameters(self):
        sd = dict(self.named_parameters())
        return [sd[name] for name, _ in ML.PARAM_SHAPES]

    def flat_parameters(self) -> torch.Tensor:
        """One contiguous fp32 buffer holding all parameters in registration order; the
        nn.Parameters are re-pointed at views of it (in-place optimizers keep it current).
        Re-flattens transparently after .to()/.cuda() moved the tensors.  Works for any network
        shape (one optimizer segment, one all-reduce range); only the standard shape has packing
        tables for the fused kernels (mlp_layout.PARAM_OFFSETS == these offsets)."""
        params = [p for _, p in self.named_parameters()]
        offsets, o = [], 0
        for p in params:
            offsets.append(o)
            o += p.numel()
        if self.is_standard():
            assert [n for n, _ in self.named_parameters()] == [n for n, _ in ML.PARAM_SHAPES]
            assert offsets == [ML.PARAM_OFFSETS[n] for n, _ in ML.PARAM_SHAPES]
        flat = self._flat
        ok = flat is not None and flat.device == params[0].device and flat.numel() == o
        if ok:
            base = flat.data_ptr()
            ok = all(p.data_ptr() == base + 4 * off for p, off in zip(params, offsets))
        if not ok:
            with torch.no_grad():
                flat = torch.cat([p.detach().reshape(-1).float() for p in params]).contiguous()
                for p, off in zip(params, offsets):
                    p.data = flat[off:off + p.numel()].view(p.shape)
            self._flat = flat
        return flat

    def attached_flat_grad(self) -> Optional[torch.Tensor]:
        """The flat gradient buffer when every parameter's .grad is a view of ONE contiguous fp32 buffer in
        parameter order (FusedAdam / FlatGradAllReduce attach them so), else None.  The backward then adds
        the weight gradients straight into it instead of returning 24 tensors for autograd to accumulate."""
        params = [p for _, p in self.named_parameters()]
        g0 = params[0].grad
        if g0 is None or g0.dtype != torch.float32 or not _capi.on_device(g0):
            return None
        base, off = g0.data_ptr(), 0
        for p in params:
            g = p.grad
            if g is None or not p.requires_grad or g.dtype != torch.float32 or not g.is_contiguous() \
                    or g.data_ptr() != base + 4 * off:
                return None
            off += p.numel()
        return torch.as_strided(g0, (off,), (1,))

    def forward(self, x):
        raise NotImplementedError(
            "scnerf_amd.NeRF is evaluated by the fused HIP kernels through render_rays / "
            "run_network (points + view directions in, raw out); a forward on pre-embedded "
            "inputs is not provided")

    def load_weights_from_keras(self, weights):
        """Same tensor order as the reference (:130-157)."""
        assert self.use_viewdirs, "Not implemented if use_viewdirs=False"
        with torch.no_grad():
            for i in range(self.D):
                self.pts_linears[i].weight.copy_(torch.from_numpy(np.transpose(weights[2 * i])))
                self.pts_linears[i].bias.copy_(torch.from_numpy(np.transpose(weights[2 * i + 1])))
            k = 2 * self.D
            self.feature_linear.weight.copy_(torch.from_numpy(np.transpose(weights[k])))
            self.feature_linear.bias.copy_(torch.from_numpy(np.transpose(weights[k + 1])))
            self.views_linears[0].weight.copy_(torch.from_numpy(np.transpose(weights[k + 2])))
            self.views_linears[0].bias.copy_(torch.from_numpy(np.transpose(weights[k + 3])))
            self.rgb_linear.weight.copy_(torch.from_numpy(np.transpose(weights[k + 4])))
            self.rgb_linear.bias.copy_(torch.from_numpy(np.transpose(weights[k + 5])))
            self.alpha_linear.weight.copy_(torch.from_numpy(np.transpose(weights[k + 6])))
            self.alpha_linear.bias.copy_(torch.from_numpy(np.transpose(weights[k + 7])))


def fix_seeds(random_seed):
    np.random.seed(random_seed)
    torch.manual_seed(random_seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(random_seed)
        torch.cuda.manual_seed_all(random_seed)
    random.seed(random_seed)
