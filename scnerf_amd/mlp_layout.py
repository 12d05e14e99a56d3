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


# ================================================================================================
# "resident" arithmetic (csrc/mlp_h3.h): the whole network on three fp16 products per product with the
# activations register-resident as pre-cut fp16 planes.  The weights stream through LDS as 16-byte MFMA
# A fragments (v_mfma_f32_32x32x16_f16: lane (n, g) holds row 32 T + n, contraction elements e = 0 .. 7 of
# the slab's lane half g) in exactly the order a wave consumes them:
#   part of output-tile PAIRS (T0, T0 + 1):  per pair, per K slab:  [Wh T0][Wh T0+1][Wl T0][Wl T0+1]
#   part of ONE output tile:                 per two K slabs:       [Wh s][Wl s][Wh s+1][Wl s+1]
# Either way four fragments (one "unit") feed six MFMAs; a 32 KB LDS chunk holds 8 units.
# Contraction element (slab sl = 2 t + u, lane half g, e) of a register-resident operand is feature
# 32 t + 16 u + 8 (e >> 2) + 4 g + (e & 3): the two 16-byte accumulator pieces (t, 2 u), (t, 2 u + 1) a lane
# owns.  Encoded points / directions: slab u, element e = PE slot 8 u + e of lane half g (pe_col).
H3_CHUNK_FRAGS = 32
H3_LAYER_FEAT, H3_LAYER_VIEWS, H3_LAYER_RGB, H3_LAYER_ALPHA = 8, 9, 10, 11
H3_SCALE_STRIDE = 8            # floats per layer in the scale table: Sw, 1/Sw, A, B, A', (3 spare)
H3_N_LAYERS = 12


def h3_feature_of(sl, g, e):
    return 32 * (sl >> 1) + 16 * (sl & 1) + 8 * (e >> 2) + 4 * g + (e & 3)


class H3Plan:
    """Fragment stream of one direction: idx [n_frags, 64, 8] int32 (flat parameter index or -1),
    meta [n_frags] uint8 (bit 0: plane 0 = h / 1 = l; bits 1..: layer slot of the scale table)."""

    def __init__(self):
        self.idx: List[np.ndarray] = []
        self.meta: List[int] = []
        self.part_units: List[Tuple[str, int]] = []

    def _frag(self, elem, T, slab, plane, layer):
        lane = np.arange(64)
        n, g = lane & 31, lane >> 5
        f = np.full((64, 8), -1, np.int32)
        for e in range(8):
            for ln in range(64):
                f[ln, e] = elem(32 * T + int(n[ln]), slab, int(g[ln]), e)
        self.idx.append(f)
        self.meta.append(plane | (layer << 1))

    def pair_part(self, name, n_tiles, slabs, elem, layer):
        assert n_tiles % 2 == 0
        for P in range(n_tiles // 2):
            for sl in slabs:
                for plane in (0, 1):
                    for T in (2 * P, 2 * P + 1):
                        self._frag(elem, T, sl, plane, layer)
        self.part_units.append((name, (n_tiles // 2) * len(slabs)))

    def single_part(self, name, T, slabs, elem, layer):
        assert len(slabs) % 2 == 0
        for sl in slabs:
            for plane in (0, 1):
                self._frag(elem, T, sl, plane, layer)
        self.part_units.append((name, len(slabs) // 2))

    def finish(self):
        n = len(self.idx)
        pad = (-n) % H3_CHUNK_FRAGS
        total = n + pad + 2 * H3_CHUNK_FRAGS            # two chunks the loader may run ahead into
        idx = np.full((total, 64, 8), -1, np.int32)
        idx[:n] = np.stack(self.idx)
        meta = np.zeros(total, np.uint8)
        meta[:n] = np.array(self.meta, np.uint8)
        return idx.reshape(-1), meta, n


def _h3_forward(lay: "Layout") -> H3Plan:
    po, IN, pd = lay.param_offsets, lay.in_pts, lay.pd
    plan = H3Plan()
    enc = [("e", u) for u in range(lay.e_slots // 8)]
    trunk = [("t", sl) for sl in range(16)]

    def kcol(slab, g, e, L, pdim, off_enc, off_feat):
        kind, s = slab
        if kind == "e":
            c = pe_col(L, 8 * s + e, g, pdim)
            return off_enc + c if c >= 0 else -1
        return off_feat + h3_feature_of(s, g, e)

    def dense(wname, ld, n_valid, L=L_PTS, pdim=pd, off_enc=0, off_feat=0):
        base = po[wname]

        def elem(o, slab, g, e):
            c = kcol(slab, g, e, L, pdim, off_enc, off_feat)
            return base + o * ld + c if (o < n_valid and c >= 0) else -1
        return elem

    plan.pair_part("L0", 8, enc, dense("pts_linears.0.weight", IN, W), 0)
    for l in range(1, 8):
        if l == 5:      # skip layer: [encoded point | h] (run_nerf_helpers.py:111-112); the encoded slabs first
            plan.pair_part("L5", 8, enc + trunk, dense("pts_linears.5.weight", W + IN, W, off_feat=IN), 5)
        else:
            plan.pair_part("L%d" % l, 8, trunk, dense("pts_linears.%d.weight" % l, W, W), l)
    plan.pair_part("LF", 8, trunk, dense("feature_linear.weight", W, W), H3_LAYER_FEAT)
    venc = [("e", u) for u in range(E_VIEWS_SLOTS // 8)]
    plan.pair_part("V", 4, trunk + venc,
                   dense("views_linears.0.weight", W + IN_VIEWS, W // 2, L=L_VIEWS, pdim=3, off_enc=W), H3_LAYER_VIEWS)
    plan.single_part("RGB", 0, [("t", sl) for sl in range(8)], dense("rgb_linear.weight", W // 2, 3), H3_LAYER_RGB)
    return plan


def _h3_backward(lay: "Layout") -> H3Plan:
    """data-gradient stream: A = W^T -- output row o is an INPUT column of the layer, the contraction runs over the
    layer's outputs (register-resident dZ)."""
    po, IN, pd = lay.param_offsets, lay.in_pts, lay.pd
    plan = H3Plan()
    trunk = list(range(16))

    def ident(limit, off=0):
        return lambda o: (off + o) if o < limit else -1

    def pe_rows(L, off=0, pdim=3, n_slots=None):
        def f(o):
            t, i = divmod(o, 32)
            r, h = row_to_rh(i)
            s = 16 * t + r
            if n_slots is not None and s >= n_slots:
                return -1
            c = pe_col(L, s, h, pdim)
            return off + c if c >= 0 else -1
        return f

    def dense_t(wname, ld, ocol, n_rows):
        base = po[wname]

        def elem(o, sl, g, e):
            col = ocol(o)
            row = h3_feature_of(sl, g, e)
            return base + row * ld + col if (col >= 0 and row < n_rows) else -1
        return elem

    def rgbt(o, sl, g, e):        # contraction over the colour channel: lane half 0, elements 0 .. 2
        return po["rgb_linear.weight"] + e * (W // 2) + o if (g == 0 and e < 3 and o < W // 2) else -1

    # (the parts that produce encoding gradients come FIRST within their layer: the part after them then starts with a
    #  complete operand and the usual pending-epilogue structure carries on)
    plan.pair_part("RGBT", 4, [0], rgbt, H3_LAYER_RGB)
    hv = list(range(8))
    plan.single_part("VTE", 0, hv, dense_t("views_linears.0.weight", W + IN_VIEWS,
                                            pe_rows(L_VIEWS, off=W, n_slots=E_VIEWS_SLOTS), W // 2), H3_LAYER_VIEWS)
    plan.pair_part("VTF", 8, hv, dense_t("views_linears.0.weight", W + IN_VIEWS, ident(W), W // 2), H3_LAYER_VIEWS)
    plan.pair_part("FT", 8, trunk, dense_t("feature_linear.weight", W, ident(W), W), H3_LAYER_FEAT)
    e_tiles = 2 if pd == 3 else 4
    for l in range(7, 0, -1):
        if l == 5:
            plan.pair_part("L5TE", e_tiles, trunk,
                           dense_t("pts_linears.5.weight", W + IN, pe_rows(L_PTS, pdim=pd, n_slots=lay.e_slots), W), 5)
            plan.pair_part("L5TH", 8, trunk, dense_t("pts_linears.5.weight", W + IN, ident(W, off=IN), W), 5)
        else:
            plan.pair_part("L%dT" % l, 8, trunk, dense_t("pts_linears.%d.weight" % l, W, ident(W), W), l)
    plan.pair_part("L0T", e_tiles, trunk,
                   dense_t("pts_linears.0.weight", IN, pe_rows(L_PTS, pdim=pd, n_slots=lay.e_slots), W), 0)
    return plan


_h3_cache: Dict[Tuple[int, str], tuple] = {}


def h3_plan(pd: int, kind: str):
    """(idx int32 [total_frags * 512], meta uint8 [total_frags], n_frags, part_units) of the forward / backward
    fragment stream of network variant pd; total_frags = n_frags rounded up to a chunk + 2 chunks of run-ahead."""
    key = (pd, kind)
    if key not in _h3_cache:
        plan = _h3_forward(layout(pd)) if kind == "fwd" else _h3_backward(layout(pd))
        idx, meta, n = plan.finish()
        _h3_cache[key] = (idx, meta, n, plan.part_units)
    return _h3_cache[key]


def h3_scale_jobs(pd: int) -> np.ndarray:
    """[12, 4] int32: (weight offset, rows, row stride = columns, bias offset) of the layers in scale-table order
    0 .. 7 trunk, feature, views, rgb, alpha."""
    lay = layout(pd)
    po = lay.param_offsets
    names = ["pts_linears.%d" % l for l in range(8)] + ["feature_linear", "views_linears.0", "rgb_linear", "alpha_linear"]
    shapes = dict(lay.param_shapes)
    jobs = []
    for n in names:
        rows, cols = shapes[n + ".weight"]
        jobs.append([po[n + ".weight"], rows, cols, po[n + ".bias"]])
    return np.array(jobs, np.int32)
