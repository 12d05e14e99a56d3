"""nerfplusplus/nerf_network.py on the fused kernels: `Embedder` describes an encoding (the encoding
itself happens inside the MLP kernels), `MLPNet` owns the parameters under the reference's names and
registration order (so state dicts and optimizer parameter indices are interchangeable)."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from .. import mlp_layout as ML


class Embedder(nn.Module):
    """(nerf_network.py:11-60) positional encoding [x, sin(x 2^k), cos(x 2^k) ...]_k.  Only the
    configuration the kernels implement is accepted: log-sampled bands up to 2^(N_freqs-1), input
    included, (sin, cos).  Calling it is not supported -- points are encoded inside the network kernel."""

    def __init__(self, input_dim, max_freq_log2, N_freqs, log_sampling=True, include_input=True,
                 periodic_fns=(torch.sin, torch.cos)):
        super().__init__()
        if not (log_sampling and include_input and len(periodic_fns) == 2 and max_freq_log2 == N_freqs - 1):
            raise NotImplementedError("the fused kernels implement log-sampled bands 2^0..2^(N-1) with the input included")
        self.input_dim = input_dim
        self.include_input = include_input
        self.N_freqs = N_freqs
        self.out_dim = input_dim + input_dim * N_freqs * 2
        self.freq_bands = (2. ** torch.linspace(0., max_freq_log2, N_freqs, device="cpu")).numpy().tolist()

    def forward(self, input):
        raise NotImplementedError("the encoding is fused into the network kernels (scnerf_mlp_fwd)")


# canonical (mlp_layout / NeRF) tensor names -> MLPNet names (nerf_network.py:86-115): identical tensors
_CANON_TO_NPP = {"alpha_linear": "sigma_layers.0", "feature_linear": "base_remap_layers.0",
                 "views_linears.0": "rgb_layers.0", "rgb_linear": "rgb_layers.2"}
for _i in range(ML.D):
    _CANON_TO_NPP["pts_linears.%d" % _i] = "base_layers.%d.0" % _i


def canonical_to_module_name(name: str) -> str:
    stem, kind = name.rsplit(".", 1)
    return _CANON_TO_NPP[stem] + "." + kind


class MLPNet(nn.Module):
    """(nerf_network.py:70-142) D ReLU layers of width W with the encoded point re-concatenated after
    layer 4, a density layer, a remap layer and the view-dependent colour head; default nn.Linear
    initialisation.  `input_ch` 63 (3-D points) or 84 (4-D background points) select the kernel variant."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_viewdirs=3, skips=[4], use_viewdirs=False):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch = input_ch
        self.input_ch_viewdirs = input_ch_viewdirs
        self.use_viewdirs = use_viewdirs
        self.skips = skips
        base = []
        dim = input_ch
        for i in range(D):
            base.append(nn.Sequential(nn.Linear(dim, W), nn.ReLU()))
            dim = W
            if i in skips and i != (D - 1):
                dim += input_ch
        self.base_layers = nn.ModuleList(base)
        self.sigma_layers = nn.Sequential(nn.Linear(dim, 1))
        self.base_remap_layers = nn.Sequential(nn.Linear(dim, 256))
        self.rgb_layers = nn.Sequential(nn.Linear(256 + input_ch_viewdirs, W // 2), nn.ReLU(),
                                        nn.Linear(W // 2, 3), nn.Sigmoid())
        self._flat: Optional[torch.Tensor] = None
        self._remap = None

    # ---- fused-kernel support ----------------------------------------------------------------
    @property
    def pt_dims(self) -> int:
        return {63: 3, 84: 4}.get(self.input_ch, 0)

    def is_standard(self) -> bool:
        return (self.D == 8 and self.W == 256 and list(self.skips) == [4] and self.use_viewdirs
                and self.pt_dims in (3, 4) and self.input_ch_viewdirs == ML.IN_VIEWS)

    def require_standard(self):
        if not self.is_standard():
            raise NotImplementedError(
                "the fused MI355X kernels cover NeRF++'s networks (D=8, W=256, skips=[4], use_viewdirs, 10 / 4 "
                "frequency bands, 3-D or 4-D points); got D=%s W=%s skips=%s use_viewdirs=%s in=%s/%s"
                % (self.D, self.W, self.skips, self.use_viewdirs, self.input_ch, self.input_ch_viewdirs))

    def flat_parameters(self) -> torch.Tensor:
        """One contiguous fp32 buffer with all parameters in registration order; the nn.Parameters become
        views of it (one optimizer segment, one all-reduce range)."""
        params = [p for _, p in self.named_parameters()]
        offsets, o = [], 0
        for p in params:
            offsets.append(o)
            o += p.numel()
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

    def module_offsets(self) -> Dict[str, int]:
        out, o = {}, 0
        for n, p in self.named_parameters():
            out[n] = o
            o += p.numel()
        return out

    def pack_remap(self):
        """(key, table) for ops.pack_weights: canonical parameter offset -> offset in this module's flat
        buffer (the reference registers the heads in another order than NeRF does)."""
        if self._remap is None:
            lay = ML.layout(self.pt_dims)
            mo = self.module_offsets()
            table = np.empty(lay.n_params, np.int64)
            for name, shape in lay.param_shapes:
                n = int(np.prod(shape))
                a = lay.param_offsets[name]
                table[a:a + n] = mo[canonical_to_module_name(name)] + np.arange(n)
            self._remap = ("npp%d" % self.pt_dims, table)
        return self._remap

    def attached_flat_grad(self) -> Optional[torch.Tensor]:
        """The flat gradient buffer when every parameter's .grad is a view of ONE contiguous fp32 buffer in registration
        order (FusedAdam / FlatGradAllReduce attach them so), else None: NerfNet's backward then adds the weight gradients
        into it with one launch per network instead of returning 24 tensors each for autograd to accumulate one by one
        (as run_nerf_helpers.NeRF.attached_flat_grad)."""
        from .. import _capi
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

    def canonical_to_module_index(self, device) -> torch.Tensor:
        """int64 [n_params] on `device`: where entry i of the kernels' flat gradient (mlp_layout order) lives in this
        module's flat buffer (pack_remap's table)."""
        key = str(device)
        if getattr(self, "_remap_dev", None) is None or self._remap_dev[0] != key:
            self._remap_dev = (key, torch.from_numpy(self.pack_remap()[1]).to(device))
        return self._remap_dev[1]

    def canonical_parameters(self):
        """the parameters in mlp_layout order (what the wgrad kernels' flat gradient is split into)"""
        sd = dict(self.named_parameters())
        return [sd[canonical_to_module_name(n)] for n, _ in ML.layout(self.pt_dims).param_shapes]

    def forward(self, input):
        raise NotImplementedError(
            "scnerf_amd MLPNet is evaluated by the fused kernels through NerfNet.forward (points + view "
            "directions in); a forward on pre-embedded inputs is not provided")
