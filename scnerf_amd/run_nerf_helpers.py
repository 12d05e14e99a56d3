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


class _EmbedFunction(torch.autograd.Function):
    """apply(x [n,d], freqs [F], include_input) -> [n, d (include_input + 2F)] (csrc/embed.hip)."""

    @staticmethod
    def forward(ctx, x, freqs, include_input):
        from . import ops
        xf = x.detach().contiguous().float()
        ctx.save_for_backward(xf, freqs)
        ctx.include_input = include_input
        return ops.embed_fwd(xf, freqs, include_input)

    @staticmethod
    def backward(ctx, g):
        from . import ops
        xf, freqs = ctx.saved_tensors
        return ops.embed_bwd(xf, g.contiguous().float(), freqs, ctx.include_input), None, None


class Embedder:
    """Positional encoding (reference :24-55): same constructor keywords, `out_dim`, and `embed(inputs)`
    -> [..., out_dim] in the reference's column order.  The render path does not call `embed`: the fused
    network kernels build the encoding in registers and only read (num_freqs, out_dim) from this object to
    check that they match what the caller configured.  Called directly it runs a stand-alone HIP kernel
    (forward and backward), for code that uses `embed_fn(x)` on its own."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d = kwargs["input_dims"]
        self.num_freqs = kwargs["num_freqs"]
        self.include_input = bool(kwargs["include_input"])
        self.out_dim = (d if self.include_input else 0) + d * 2 * self.num_freqs
        fns = list(kwargs.get("periodic_fns", [torch.sin, torch.cos]))
        self._sin_cos = len(fns) == 2 and fns[0] is torch.sin and fns[1] is torch.cos
        max_freq = kwargs["max_freq_log2"]
        if kwargs.get("log_sampling", True):                   # the reference's freq_bands, computed the same way
            self.freq_bands = 2. ** torch.linspace(0., max_freq, steps=self.num_freqs)
        else:
            self.freq_bands = torch.linspace(2. ** 0., 2. ** max_freq, steps=self.num_freqs)
        self._bands_on = {}

    def embed(self, inputs):
        if not self._sin_cos:
            raise NotImplementedError("the embedding kernel implements periodic_fns = [torch.sin, torch.cos]")
        if not _capi.on_device(inputs):
            raise RuntimeError("inputs must be on the GPU: scnerf_amd has no CPU path")
        key = str(inputs.device)
        if key not in self._bands_on:
            self._bands_on[key] = self.freq_bands.float().to(inputs.device)
        lead = inputs.shape[:-1]
        out = _EmbedFunction.apply(inputs.reshape(-1, inputs.shape[-1]), self._bands_on[key], self.include_input)
        return out.reshape(*lead, self.out_dim)

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
        """x [..., input_ch + input_ch_views] = [encoded point | encoded view direction] -> [..., 4] (rgb logits,
        sigma) as the reference's forward (:105-128) for the use_viewdirs network.

        The fused kernels encode points themselves, so this entry takes the RAW point and direction out of the
        encodings' leading columns (include_input puts them at [0:3] and [input_ch : input_ch + 3]) and evaluates
        the network on those -- exact whenever `x` IS the positional encoding of its own leading columns, which is
        how every caller in the reference builds it (create_nerf.py:18-32); that is checked on sixteen rows of EVERY
        call (two small reductions: this entry is off the render path, which never materialises encodings), so a
        perturbed or partly zeroed encoding raises instead of silently returning the result for other inputs.
        Gradient: d x is placed on those leading columns ONLY (it already contains the encoding's chain rule), so
        embed -> forward differentiates correctly end to end, while x.grad of the sin / cos columns stays zero."""
        if not self.is_standard():
            return self._forward_layer_by_layer(x)
        if x.shape[-1] != self.input_ch + self.input_ch_views:
            raise ValueError("expected %d columns, got %d" % (self.input_ch + self.input_ch_views, x.shape[-1]))
        lead = x.shape[:-1]
        flat = x.reshape(-1, x.shape[-1])
        pts, views = flat[:, 0:3], flat[:, self.input_ch:self.input_ch + 3]
        if flat.shape[0] > 0:
            rows = flat[:: max(1, flat.shape[0] // 16)][:16].detach()
            want_p = get_embedder(ML.L_PTS, 0)[0](rows[:, 0:3].contiguous())
            want_v = get_embedder(ML.L_VIEWS, 0)[0](rows[:, self.input_ch:self.input_ch + 3].contiguous())
            err = max(float((want_p - rows[:, :self.input_ch]).abs().max()),
                      float((want_v - rows[:, self.input_ch:]).abs().max()))
            if not err <= 1e-3:
                raise ValueError("NeRF.forward: x is not the positional encoding (multires %d / %d, include_input) of "
                                 "its own leading columns (max deviation %g)" % (ML.L_PTS, ML.L_VIEWS, err))
        from .create_nerf import _QueryFunction
        raw = _QueryFunction.apply(pts.reshape(-1, 1, 3), views, self, torch.is_grad_enabled(), *self.ordered_parameters())
        return raw.reshape(*lead, 4)

    def _forward_layer_by_layer(self, x):
        """Any other shape (D, W, skips, no view directions, another encoding width): the reference's forward as it stands
        (:105-128), layer by layer on the device's GEMM library -- `x` really is [encoded point | encoded direction] here.
        The fused kernels exist for the one shape every SCNeRF configuration uses; render_rays takes this route through the
        caller's `network_query_fn` (render.render_rays: the opaque-callable path) with the sampling, search and compositing
        still on the HIP kernels."""
        import torch.nn.functional as F
        input_pts, input_views = torch.split(x, [self.input_ch, self.input_ch_views], dim=-1)
        h = input_pts
        for i, layer in enumerate(self.pts_linears):
            h = F.relu(layer(h))
            if i in self.skips:
                h = torch.cat([input_pts, h], -1)
        if self.use_viewdirs:
            alpha = self.alpha_linear(h)
            feature = self.feature_linear(h)
            h = torch.cat([feature, input_views], -1)
            for layer in self.views_linears:
                h = F.relu(layer(h))
            return torch.cat([self.rgb_linear(h), alpha], -1)
        return self.output_linear(h)

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
