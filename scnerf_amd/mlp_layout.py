"""Host-side description of how the NeRF MLP's weights are laid out for the fused MFMA
kernels (scnerf_amd/csrc/mlp_fwd.hip, mlp_bwd.hip).

Design recap (DESIGN.md section "MLP kernels"): a wave owns 32 samples and keeps the whole
activation vector of those samples in registers, in the accumulator layout of
v_mfma_f32_32x32x2_f32 computed *transposed* (D[feature][sample]): lane (m, h) = (l & 31,
l >> 5) holds, for sample m, the features  feat_of(t, r, h) = 32 t + (r & 3) + 8 (r >> 2) + 4 h
of n-tile t in accumulator register r.  Because the contraction order of a dot product is
free, those same registers are fed straight back as the B operand of the next layer: MFMA
step s = 16 t + r contracts features (feat_of(t,r,0), feat_of(t,r,1)).  Only the weights
move: they are pre-gathered ("packed") into the exact order the A operand is consumed in,
streamed L2 -> LDS in chunks and read with ds_read_b128 (4 consecutive steps per lane).

The packing is a pure gather from the flat parameter buffer, so it is expressed as int32
index tables built here once per model; the device side is one gather kernel
(scnerf_gather_f32) run whenever the weights changed.

Positional-encoding inputs use "slots": slot s of lane-half h is one embedding column
(pe_col) chosen so that a lane needs one sincos per angle.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

W = 256
D = 8
SKIP = 4
L_PTS = 10
L_VIEWS = 4
IN_VIEWS = 3 + 6 * L_VIEWS    # 27
E_VIEWS_SLOTS = 16            # steps of the encoded-view part (K = 32 incl. 5 pads)


def feat_of(t, r, h):
    return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h


def row_to_rh(i):
    """inverse of feat_of within a tile: row i (0..31) -> (r, h)."""
    return (i & 3) + 4 * (i >> 3), (i >> 2) & 1


def pe_col(L: int, s: int, h: int, pd: int = 3) -> int:
    """Embedding column (torch order [x, sin f0, cos f0, sin f1, ...], run_nerf_helpers.py:33-55 /
    nerfplusplus/nerf_network.py:41-60) carried by PE slot s on lane-half h; -1 = zero pad.
    pd = 3: half h owns x|y and shares z; pd = 4: half h owns (x|y, z|w)."""
    if pd == 3:
        if s < 3 * L:
            f, j = divmod(s, 3)
            if j == 0:
                return 3 + 6 * f + h               # sin of x (h=0) / y (h=1)
            if j == 1:
                return 3 + 6 * f + 3 + h           # cos of x / y
            return 3 + 6 * f + (2 if h == 0 else 5)  # z: sin on h=0, cos on h=1
        if s == 3 * L:
            return h                               # raw x / y
        if s == 3 * L + 1:
            return 2 if h == 0 else -1             # raw z / pad
        return -1
    if s < 4 * L:
        f, j = divmod(s, 4)
        coord = (0 if j < 2 else 2) + h
        return 4 + 8 * f + (4 if (j & 1) else 0) + coord
    if s == 4 * L:
        return h
    if s == 4 * L + 1:
        return 2 + h
    return -1


@dataclass
class Part:
    name: str
    nstep: int
    nt: int
    cs: int                  # steps per LDS chunk

    @property
    def floats(self):
        return self.nstep * self.nt * 64

    @property
    def chunk_floats(self):
        return self.cs * self.nt * 64


def _part_index(part: Part, src) -> np.ndarray:
    """index array of one part; src(tile, row_i, step, half) -> flat param index or -1.
    Layout: [chunk][tile][group of 4 steps][lane][4]."""
    nc = part.nstep // part.cs
    c, t, g, lane, j = np.meshgrid(np.arange(nc), np.arange(part.nt), np.arange(part.cs // 4),
                                   np.arange(64), np.arange(4), indexing="ij")
    step = c * part.cs + 4 * g + j
    half = lane >> 5
    i = lane & 31
    f = np.vectorize(src, otypes=[np.int64])
    return f(t, i, step, half).reshape(-1).astype(np.int32)


def _feat_step(step, half):
    return feat_of(step // 16, step % 16, half)


def lane_vector_table(base: int, n_valid: int, ntiles: int) -> np.ndarray:
    """Per-feature vectors (biases, the density head's weights) in the order a lane loads them: entry
    ((4 t + q) * 2 + h) * 4 + j holds feature feat_of(t, 4 q + j, h) -- one 16-byte load per (tile, quarter)
    with the lane half h as part of the address, no scalar loads + per-register selects."""
    idx = np.full(ntiles * 16 * 2, -1, np.int32)
    for t in range(ntiles):
        for q in range(4):
            for h in range(2):
                for j in range(4):
                    n = feat_of(t, 4 * q + j, h)
                    if n < n_valid:
                        idx[((4 * t + q) * 2 + h) * 4 + j] = base + n
    return idx


MASK_WORDS_PER_SAMPLE = 9 * 8     # lane-native ReLU bit masks behind the row sections
GRAD_SECTIONS = [("dz%d" % l, W) for l in range(D)] + [("dfeat", W), ("dzv", W // 2)]
GRAD_FLOATS_PER_SAMPLE = sum(w for _, w in GRAD_SECTIONS)      # 2432
TILED_SECTIONS = {"feat", "hv", "dfeat", "dzv"} | {"act%d" % l for l in range(D)} | {"dz%d" % l for l in range(D)}


class Layout:
    """Everything that depends on the network variant: pd = 3, points (x, y, z) -> 63 encoded columns
    (the SCNeRF networks, NeRF++'s foreground net); pd = 4, points (x, y, z, 1/r) -> 84 columns
    (NeRF++'s background net, nerfplusplus/ddp_model.py:62-71).  Mirrors csrc/mlp_common.h Var<PD>."""

    def __init__(self, pd: int):
        assert pd in (3, 4)
        self.pd = pd
        self.in_pts = pd + 2 * pd * L_PTS              # 63 / 84
        self.e_slots = 32 if pd == 3 else 48           # MFMA steps of the encoded-point parts
        self.e_width = 64 if pd == 3 else 128          # row width of the saved encodings
        self.e_tiles = 2 if pd == 3 else 4             # dgrad tiles of the d-encoding parts (4th = zero pad)
        self.e_cs = 64 if pd == 3 else 32
        IN = self.in_pts
        # reference NeRF registration order (NeRF/run_nerf_helpers.py:88-103) == flat buffer order
        shapes: List[Tuple[str, Tuple[int, ...]]] = []
        for i in range(D):
            fan_in = IN if i == 0 else (W + IN if (i - 1) == SKIP else W)
            shapes.append(("pts_linears.%d.weight" % i, (W, fan_in)))
            shapes.append(("pts_linears.%d.bias" % i, (W,)))
        shapes += [
            ("views_linears.0.weight", (W // 2, IN_VIEWS + W)), ("views_linears.0.bias", (W // 2,)),
            ("feature_linear.weight", (W, W)), ("feature_linear.bias", (W,)),
            ("alpha_linear.weight", (1, W)), ("alpha_linear.bias", (1,)),
            ("rgb_linear.weight", (3, W // 2)), ("rgb_linear.bias", (3,)),
        ]
        self.param_shapes = shapes
        self.param_offsets: Dict[str, int] = {}
        o = 0
        for n, sh in shapes:
            self.param_offsets[n] = o
            o += int(np.prod(sh))
        self.n_params = o                              # 595 844 / 606 596
        ES = self.e_slots
        self.fwd_parts = ([Part("E0", ES, 8, 16)] + [Part("H%d" % l, 128, 8, 16) for l in (1, 2, 3, 4)]
                          + [Part("E5", ES, 8, 16), Part("H5", 128, 8, 16), Part("H6", 128, 8, 16),
                             Part("H7", 128, 8, 16), Part("HF", 128, 8, 16), Part("VF", 128, 4, 32),
                             Part("VE", 16, 4, 16), Part("RGB", 64, 1, 64)])
        self.bwd_parts = ([Part("RGBT", 4, 4, 4), Part("VTF", 64, 8, 16), Part("VTE", 64, 1, 64),
                           Part("FT", 128, 8, 16), Part("L7T", 128, 8, 16), Part("L6T", 128, 8, 16),
                           Part("L5TH", 128, 8, 16), Part("L5TE", 128, self.e_tiles, self.e_cs)]
                          + [Part("L%dT" % l, 128, 8, 16) for l in (4, 3, 2, 1)]
                          + [Part("L0T", 128, self.e_tiles, self.e_cs)])
        self.fwd_stream = sum(p.floats for p in self.fwd_parts)
        self.bwd_stream = sum(p.floats for p in self.bwd_parts)
        # tail sections of the packed forward buffer (lane-vector layout, see lane_vector_table)
        self.fwd_bias = self.fwd_stream                 # 8 trunk layers x 256
        self.fwd_bias_f = self.fwd_bias + 8 * 256
        self.fwd_bias_v = self.fwd_bias_f + 256
        self.fwd_bias_rgb = self.fwd_bias_v + 128
        self.fwd_alpha_w = self.fwd_bias_rgb + 32
        self.fwd_alpha_b = self.fwd_alpha_w + 256
        self.fwd_total = self.fwd_alpha_b + 4
        self.bwd_alpha_w = self.bwd_stream
        self.bwd_total = self.bwd_alpha_w + 256
        # activation workspace saved by the training forward: wide sections tile-native, the encodings
        # row-major [P][width]; epts last (the only variant-dependent width)
        self.save_sections = [("act%d" % l, W) for l in range(D)] + [("feat", W), ("hv", W // 2), ("eviews", 32),
                                                                     ("epts", self.e_width)]
        self.save_floats_per_sample = sum(w for _, w in self.save_sections)
        self._cache: Dict[str, np.ndarray] = {}

    # ---- workspace sizes ----
    def save_floats(self, P: int) -> int:
        return (self.save_floats_per_sample + MASK_WORDS_PER_SAMPLE) * padded_samples(P)

    # ---- index tables ----
    def forward_index(self) -> np.ndarray:
        if "f" not in self._cache:
            self._cache["f"] = self._build_forward_index()
        return self._cache["f"]

    def backward_index(self) -> np.ndarray:
        if "b" not in self._cache:
            self._cache["b"] = self._build_backward_index()
        return self._cache["b"]

    def _build_forward_index(self) -> np.ndarray:
        po, IN, pd = self.param_offsets, self.in_pts, self.pd
        out = []

        def dense(wname, ld, n_valid, kcol):
            base = po[wname]

            def src(t, i, step, half):
                n = 32 * t + i
                col = kcol(step, half)
                return base + n * ld + col if (n < n_valid and col >= 0) else -1
            return src

        for p in self.fwd_parts:
            if p.name == "E0":
                s = dense("pts_linears.0.weight", IN, W, lambda st, h: pe_col(L_PTS, st, h, pd))
            elif p.name == "E5":       # skip layer: encoded points are the first columns (helpers.py:111-112)
                s = dense("pts_linears.5.weight", W + IN, W, lambda st, h: pe_col(L_PTS, st, h, pd))
            elif p.name == "H5":
                s = dense("pts_linears.5.weight", W + IN, W, lambda st, h: IN + _feat_step(st, h))
            elif p.name.startswith("H") and p.name != "HF":
                s = dense("pts_linears.%s.weight" % p.name[1:], W, W, _feat_step)
            elif p.name == "HF":
                s = dense("feature_linear.weight", W, W, _feat_step)
            elif p.name == "VF":       # views layer input = [feature(256), encoded dir(27)] (:117)
                s = dense("views_linears.0.weight", W + IN_VIEWS, W // 2, _feat_step)
            elif p.name == "VE":
                def kc(st, h):
                    c = pe_col(L_VIEWS, st, h)
                    return W + c if c >= 0 else -1
                s = dense("views_linears.0.weight", W + IN_VIEWS, W // 2, kc)
            elif p.name == "RGB":
                s = dense("rgb_linear.weight", W // 2, 3, _feat_step)
            else:
                raise AssertionError(p.name)
            out.append(_part_index(p, s))

        def halfpair(bname, n_valid, ntiles):
            return lane_vector_table(po[bname], n_valid, ntiles)

        for l in range(D):
            out.append(halfpair("pts_linears.%d.bias" % l, W, 8))
        out.append(halfpair("feature_linear.bias", W, 8))
        out.append(halfpair("views_linears.0.bias", W // 2, 4))
        out.append(halfpair("rgb_linear.bias", 3, 1))
        out.append(halfpair("alpha_linear.weight", W, 8))
        out.append(np.array([po["alpha_linear.bias"], -1, -1, -1], np.int32))
        idx = np.concatenate(out)
        assert idx.shape[0] == self.fwd_total, (idx.shape, self.fwd_total)
        return idx

    def _build_backward_index(self) -> np.ndarray:
        """dgrad stream: the A operand is W^T -- row i of tile t is an *input* column of the
        layer, the contraction runs over the layer's outputs (held in registers as dZ)."""
        po, IN, pd = self.param_offsets, self.in_pts, self.pd
        out = []

        def dense_t(wname, ld, ocol, krow):
            base = po[wname]

            def src(t, i, step, half):
                col = ocol(t, i)
                row = krow(step, half)
                return base + row * ld + col if (col >= 0 and row >= 0) else -1
            return src

        def ident(limit, off=0):
            return lambda t, i: (off + 32 * t + i) if (32 * t + i) < limit else -1

        def pe_rows(L, t_base, off=0, pdim=3):
            def f(t, i):
                r, h = row_to_rh(i)
                c = pe_col(L, 16 * (t - t_base) + r, h, pdim)
                return off + c if c >= 0 else -1
            return f

        for p in self.bwd_parts:
            if p.name == "RGBT":        # out: hv features (128); contraction: rgb channel 2s+h (<3)
                s = dense_t("rgb_linear.weight", W // 2, ident(W // 2),
                            lambda st, h: (2 * st + h) if (2 * st + h) < 3 else -1)
            elif p.name == "VTF":       # out: feature (views-layer input cols 0..255); contraction: hv (128)
                s = dense_t("views_linears.0.weight", W + IN_VIEWS, ident(W), _feat_step)
            elif p.name == "VTE":       # out: encoded view-direction slots (cols 256..282)
                s = dense_t("views_linears.0.weight", W + IN_VIEWS, pe_rows(L_VIEWS, 0, off=W), _feat_step)
            elif p.name == "FT":
                s = dense_t("feature_linear.weight", W, ident(W), _feat_step)
            elif p.name == "L5TH":      # out: h (the columns after the encoded point of the skip layer)
                s = dense_t("pts_linears.5.weight", W + IN, ident(W, off=IN), _feat_step)
            elif p.name == "L5TE":      # out: encoded point slots (the first columns)
                s = dense_t("pts_linears.5.weight", W + IN, pe_rows(L_PTS, 0, pdim=pd), _feat_step)
            elif p.name == "L0T":
                s = dense_t("pts_linears.0.weight", IN, pe_rows(L_PTS, 0, pdim=pd), _feat_step)
            else:
                l = int(p.name[1])
                s = dense_t("pts_linears.%d.weight" % l, W, ident(W), _feat_step)
            out.append(_part_index(p, s))
        out.append(lane_vector_table(po["alpha_linear.weight"], W, 8))
        idx = np.concatenate(out)
        assert idx.shape[0] == self.bwd_total
        return idx


_layouts: Dict[int, Layout] = {}


def layout(pd: int = 3) -> Layout:
    if pd not in _layouts:
        _layouts[pd] = Layout(pd)
    return _layouts[pd]


def padded_samples(P: int) -> int:
    return (P + 127) // 128 * 128


def grad_floats(P: int) -> int:
    return GRAD_FLOATS_PER_SAMPLE * padded_samples(P)


def untile(block, width: int, P: int):
    """tile-native section (flat array of width * padded P floats) -> row-major [P, width]:
    per 32 samples a block [t][q][lane = m + 32 h][j] holding feature 32 t + 8 q + 4 h + j."""
    Pp = block.shape[0] // width
    a = block.reshape(Pp // 32, width // 32, 4, 2, 32, 4)          # tile, t, q, h, m, j
    a = a.transpose(0, 4, 1, 2, 3, 5).reshape(Pp, width)           # tile, m, t, q, h, j
    return a[:P]


def section_offsets(sections, P):
    """offsets (floats) of the workspace sections: every section spans width * padded_samples(P)."""
    Pp = padded_samples(P)
    off, out = 0, {}
    for name, w in sections:
        out[name] = off
        off += w * Pp
    return out, off


# ---- the standard (pd = 3) network under the names the rest of the package uses ----------------
_STD = layout(3)
IN_PTS = _STD.in_pts          # 63
E_PTS_SLOTS = _STD.e_slots
PARAM_SHAPES = _STD.param_shapes
PARAM_OFFSETS = _STD.param_offsets
N_PARAMS = _STD.n_params      # 595 844
FWD_PARTS, BWD_PARTS = _STD.fwd_parts, _STD.bwd_parts
FWD_STREAM, BWD_STREAM = _STD.fwd_stream, _STD.bwd_stream
FWD_BIAS, FWD_BIAS_F, FWD_BIAS_V, FWD_BIAS_RGB = _STD.fwd_bias, _STD.fwd_bias_f, _STD.fwd_bias_v, _STD.fwd_bias_rgb
FWD_ALPHA_W, FWD_ALPHA_B, FWD_TOTAL = _STD.fwd_alpha_w, _STD.fwd_alpha_b, _STD.fwd_total
BWD_ALPHA_W, BWD_TOTAL = _STD.bwd_alpha_w, _STD.bwd_total
SAVE_SECTIONS = _STD.save_sections
SAVE_FLOATS_PER_SAMPLE = _STD.save_floats_per_sample      # 2592 -> 10 368 B / sample


def forward_index() -> np.ndarray:
    return _STD.forward_index()


def backward_index() -> np.ndarray:
    return _STD.backward_index()


def save_floats(P: int) -> int:
    return _STD.save_floats(P)
