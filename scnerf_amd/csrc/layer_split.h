// layer_split.h -- one 256 -> 256 linear layer of the network over ALL samples on the bf16 matrix pipe at fp32
// accuracy:      Z[p][n] = act( sum_k W[n][k] X[p][k] + b[n] )          (tile-native X, Z of width 256)
//
// Arithmetic as wgrad256_split.h: every fp32 number is cut exactly into three bf16 numbers (h, m, l) and a
// product is the fp32 sum of six bf16 x bf16 partial products -- (Wh Xh)(Wh Xm)(Wh Xl)(Wm Xh)(Wm Xm)(Wl Xh).
// The weights are cut ONCE per optimizer step by the packing pass; only the activations are cut here.
//
// kHalf3 instantiations: the same kernel on THREE products per product -- every operand scaled by a power of two and
// cut into two fp16 numbers, (Wh Xh)(Wh Xl)(Wl Xh) on v_mfma_f32_32x32x16_f16; scales per sample (activations,
// gradients: a sample is a column of the transposed product, so the scale comes back out lane-wise in the epilogue)
// and per layer (weights); see Layer::amax_in / amax_out.  Error against fp64 as for the six bf16 products (the dropped
// terms are 2^-22 of the product in both), half the MFMA work: 0.355-0.38 ms per layer against 0.48-0.51.
//
// Shape.  The product is computed transposed, D[feature][sample], so an accumulator tile is the tile-native
// piece layout of mlp_common.h (lane (m, h) owns features 32 t + 8 q + 4 h + j of sample m: one 16-byte store
// per (t, q)).  A workgroup = 2 x 2 waves; wave (wn, wp) owns features 128 wn .. +127 of samples 128 wp .. +127 of a
// 256-sample block: 4 x 4 accumulator tiles = 256 AGPRs, the largest tile a wave can hold.
//   A operand (weights): the three planes of a 16-wide K slab arrive from L2 in fragment order
//     [slab][plane][feature tile T][lane][8 bf16]  (lane = (row n & 31, k-group g): W[32 T + n][16 s + 8 g .. +7])
//     and are copied as they are into one of two 24 KB LDS buffers; a wave reads its 4 tiles x 3 planes with
//     ds_read_b128, ONCE per slab (48 VGPRs) for all four sample tiles.
//   B operand (activations): NOT through LDS.  The 8 consecutive k a lane needs are the two 16-byte pieces
//     (t, q, m + 32 h'), h' = 0, 1, of its sample in the tile-native block: two coalesced global loads, then the
//     cut in registers (and / sub / and / sub per float, v_perm per pair and plane).  Both waves of a wp pair do
//     this for the same samples (the second one hits in L1).
// A slab of 16 k = 4 sample tiles x 6 products x 4 feature tiles = 96 MFMAs; the fillers (the cut of the NEXT
// sample tile's fragment, the loads two slabs ahead, the weight copy one slab ahead, the weight fragment reads
// as their registers fall free) sit between the MFMAs, one memory instruction every 6 slots; one barrier per slab.
//
// Measured (tools/ubench/layer_split_lab.hip, profiles/r02f_layer_split_lab.txt; P = 786 432): 0.51 ms per layer as a
// launch of its own, 0.48-0.49 in a chain of eight (0.80 ms for a layer inside the fused fp32-MFMA kernel).
// What bounds it.  The lab reads the shader clock inside the kernel (s_memtime against the 100 MHz s_memrealtime):
//     MFMAs + weight stream, activation planes never written   615 k cycles per workgroup at 2.03-2.09 GHz  0.30-0.32 ms
//     the same with pseudo-random bits in the planes           627 k cycles             at 1.68-1.70 GHz  0.38-0.39 ms
//     + activation loads and cuts (no epilogue)                720 k cycles             at 1.69-1.72 GHz  0.44 ms
//     whole kernel                                             842 k cycles             at 1.71-1.81 GHz  0.51 ms
// against 589 824 cycles of MFMA work per workgroup: the matrix pipe is busy 70 % of the cycles (82 % between the
// epilogues), and the chip runs this kernel at 1.7 GHz -- it is POWER-bound like wgrad256_split.h, and what the
// matrix pipe draws depends on the operand bits (static registers: 2.05 GHz).  Two earlier readings of this kernel
// were wrong: "busy 46-55 % at 2.04 GHz, stall-bound" came from GRBM_GUI_ACTIVE / duration, which is not the shader
// clock (it reads 10 GHz on a 3 us copy kernel); and "the activation loads cost 0.12 ms, unexplained" compared runs
// with and without real operand bits -- half of that difference is the clock, not a stall.  Ruled out on the way
// (lab flags below, none changed the time): HBM locality (chains in groups of 2 blocks read what they wrote 40 us ago:
// same time), two waves loading the same lines (kNoDupX), all workgroups walking K in step (kRotate), the memory
// system altogether (kXfromW: the loads fetch L2-resident weight pieces instead, 0.425 vs 0.447 ms).
// What did cost cycles, found in the ISA and fixed in round 2:
//   * a bias read from global memory inside the epilogue waits on vmcnt, which also counts the stores just issued --
//     one store round trip per 16-byte piece, 20 us per block (table in LDS);
//   * then the LDS read itself: issued right where it was used, each of the 64 pieces per wave and block waited out an
//     LDS round trip (176 k -> 122 k cycles per workgroup with the read one piece ahead and shared by the four sample
//     tiles);
//   * 64-bit per-lane pointers: ~20 VALU of address arithmetic per load, lane-invariant parts hoisted out of the block
//     loop into registers that spill, scratch reloads waiting on vmcnt(0) in the middle of the stores (wave-uniform
//     bases made opaque with an empty asm + 32-bit lane offsets, pinned_here for what must not be hoisted);
//   * ~50 scalar instructions per load (64-bit tile arithmetic, the layer's pointers re-fetched from the argument
//     block behind s_waitcnt lgkmcnt(0)): scalar instructions issue from the same in-order stream as the MFMAs
//     (addresses once per slab, ahead_of);
//   * memory instructions issued back to back (six weight pieces in six slots, two activation pieces in one) by four
//     lock-stepped waves: 0.344 -> 0.317 ms for MFMAs + weight stream with one every 6 slots;
//   * accumulator reads hoisted out of their pieces spill; `if (wave == k)` around a load makes the wait-count pass
//     fall back to vmcnt(0) behind the branch (0.77 ms); buffer-descriptor loads bloat the code through unswitching.
// Tried, no effect (consistent with a power bound -- the instructions cost the same wherever they sit): spreading the
// epilogue over the MFMA slots around the block seam, staggering the workgroups, non-temporal loads, no barriers
// (racy), four waves side by side along the samples.  Run on half of the CUs each workgroup is 21-25 % faster.
#pragma once
#include <type_traits>

#include <scn_wave.h>

namespace scn {
namespace lsp {

constexpr int kThreads = 256;
constexpr int kSlabs = 16;                          // K = 256 in slabs of 16
constexpr int kSlabUnits = 3 * 8 * 64;              // 16-byte fragments of one slab image
constexpr int kSlabShorts = kSlabUnits * 8;
constexpr int kAmaxSlots = 3;                       // blocks of a group whose maxima go from layer to layer through LDS
// two slab images + the bias table + the per-sample maxima of the group's blocks [slot][feature half][256 samples]
constexpr unsigned kLdsBytes = 2u * kSlabUnits * 16u + 1024u + kAmaxSlots * 2u * 256u * 4u;

struct Layer {
    const float* X;        // tile-native, width 256: K slabs 0 .. 15
    const float* X2;       // row-major [Ppad][x2_ld] (the skip layer's encoded points): K slabs 16 .. n_k - 1 (else = X)
    int n_k;               // K slabs per block: 16, or 16 + x2 width / 16 (even)
    int relu;
    const short* W;        // [n_k][3][8][64][8] bf16 bits (pack_planes)
    const float* bias;     // lane-vector table of 8 tiles: entry ((4 t + q) * 2 + h) * 4 + j
    float* Z;              // tile-native, width 256
    unsigned* mask;        // ReLU bits [wave tile][64 lanes][4 words] or nullptr (mode 0)
    const unsigned* mask_in;   // mode 1: the gate, [wave tile][64 lanes][4 words]
    const float* vec;          // mode 1: per-sample scalar of the rank-1 term, element p * vec_stride (p < n_vec), or nullptr
    // Two-way fp16 cut (kHalf3 instantiations).  fp16 has five exponent bits: every operand carries a power-of-two
    // scale.  A sample is a COLUMN of the transposed product, so activations / gradients are scaled per sample:
    // the producing layer's epilogue leaves the maximum |value| of each sample over either half of the features
    // (amax_out, [2][Ppad]), the consumer derives 2^k from the two halves (amax_in) while cutting and divides it out
    // again, lane-wise, in its own epilogue; the weights carry one scale per layer from the packing pass.
    const float* amax_in;      // kHalf3: [2][Ppad] maxima of X (nullptr: Args::x_scale for every sample -- the lab)
    float* amax_out;           // any instantiation: [2][Ppad] maxima of Z, or nullptr
    const float* w_inv_scale;  // kHalf3: 1 / (the power of two the packer multiplied this layer's weights by)
};
// A chain of layers in ONE launch: layer l + 1 of a 256-sample block reads what layer l wrote for the SAME block, and
// a workgroup keeps its blocks from layer to layer, so there is no dependency between workgroups -- the software
// pipeline simply runs on across the layer boundary.  A workgroup takes a GROUP of its blocks (>= 2) through all the
// layers before it turns to the next group: the loads two slabs ahead then fetch the group's first block in the next
// layer, stored a whole block (~40 us) earlier.  Groups of 2 keep what a layer reads within 128 MB of what the chip
// wrote last (with default-policy stores: 2-3 % faster than layer by layer over all blocks, lab; the kernel is not
// bound by where its activations come from).  The launcher only chains when every workgroup owns at least two
// blocks.  Saves the launch gaps and pipeline refills of eight launches per pass: 0.51 -> 0.48-0.49 ms per layer.
constexpr int kMaxLayers = 8;
struct Args {
    Layer layer[kMaxLayers];
    int n_layers;
    int group;             // blocks a workgroup takes through ALL layers before it turns to its next blocks (>= 2;
                           // >= its block count: layer by layer over all of them)
    int x2_ld;
    long Ppad;             // samples covered (multiple of 128)
    // mode 0: Z = act(W X + b), ReLU bits written; mode 1 (data gradients): Z = select(mask_in, W X + bias[n] * vec[p])
    // -- the ReLU gate of the layer below and the rank-1 density-head term (bias = alpha_linear's weights, vec = d sigma)
    int mode;
    int vec_stride;
    long n_vec;
    unsigned long long* clock_probe;   // lab only (kClockProbe): [workgroup][2]
    float x_scale, out_scale;          // lab only (kHalf3)
};

enum : int {
    kNoEpilogue = 1,      // timing experiment: only the last block is stored
    kNoCut = 2,           // timing experiment: no loads / cuts of X (planes hold garbage)
    kNoCutMath = 8,       // timing experiment: X is loaded but not cut (the raw words serve as planes)
    kSameX = 32,          // timing experiment: every activation load fetches the tiles of block 0 (always an L2 hit)
    kNoWCopy = 128,       // timing experiment: the weight slabs are not copied (LDS holds the first two)
    kRotate = 512,        // timing experiment: every block starts its K loop at another slab (X only: wrong results)
    kXfromW = 1024,       // timing experiment: the activation loads fetch pieces of the weight slab instead (wrong results)
    kRandomX = 2048,      // timing experiment (with kNoCut): the activation planes hold pseudo-random bits instead of garbage
    kClockProbe = 4096,   // lab: every workgroup writes its shader-cycle and reference-tick counts to Args::clock_probe
    kHalf3 = 8192,        // TWO fp16 planes per operand and three products (Wh Xh)(Wh Xl)(Wl Xh); the operands scaled by
                          // powers of two (Layer::amax_in / w_inv_scale; in the lab Args::x_scale / out_scale)
    kAmaxOut = 16384,     // the epilogue also leaves the per-sample maxima of what it stores (Layer::amax_out)
    kNoDupX = 256,        // timing experiment: the two waves that share samples load DIFFERENT tiles (wrong results)
    kNoBarrier = 64,      // timing experiment: no workgroup barriers in the slab loop (racy: results are wrong)
    kPlainStore = 4,      // default-policy stores instead of non-temporal ones: what a chain in groups of 2 blocks uses
                          // (the next layer re-reads the block 40 us later: 2-3 % faster); alone no difference
    kNoZStore = 16,       // experiment: the epilogue computes but stores only the mask words
};

template <int N> using I = std::integral_constant<int, N>;

template <int FLAGS>
__global__ __launch_bounds__(kThreads, 1) void layer_split_kernel(Args a) {
    short* lds = dynamic_lds<short>();
    float* const lds_bias = reinterpret_cast<float*>(lds + 2 * kSlabShorts);
    float* const lds_amax = lds_bias + 256;
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const int wn = wave >> 1, wp = wave & 1;
    const int m = lane & 31, g = lane >> 5;
    const long n_tiles = a.Ppad / 32;
    const long n_blocks = (n_tiles + 7) / 8;
    if ((long)blockIdx.x >= n_blocks) return;
    const int my_blocks = (int)((n_blocks - blockIdx.x + gridDim.x - 1) / gridDim.x);
    const int n_layers = a.n_layers;

    f32x16 acc[4][4];       // written by MFMAs only: a block's first product starts from the constant 0

    // ---- X: lane (m, g) of sample tile j, slab s: pieces ((2 s + g) * 64 + m + 32 h') * 4 floats, h' = 0, 1
    // Every load is `global_load v, v_off, s[base]`: a wave-uniform base (uniform_global: opaque to the compiler) + a
    // 32-bit per-lane byte offset.  With 64-bit per-lane pointers each load cost ~20 VALU of address arithmetic and the
    // compiler recycled landed staging registers for the temporaries behind an `s_waitcnt vmcnt(0)`.
    const int wp_u = uniform(wp);
    const unsigned xoff_a = (unsigned)((g * 64 + m) * 16);                                   // main operand, piece h' = 0
    const unsigned xoff_b = (unsigned)((m * a.x2_ld + 8 * g) * 4);                           // second operand
    // The addresses of a slab's loads are wave-uniform and computed ONCE per slab (ahead_of): scalar instructions issue
    // from the same in-order stream as the MFMAs, and ~50 of them per load -- 64-bit tile arithmetic, the layer's
    // pointers fetched from the argument block behind an `s_waitcnt lgkmcnt(0)` -- was what the activation loads cost
    // (0.05-0.08 ms per layer even when every load hit in L1).  A load now adds j * stride to the slab's base.
    struct LayerPtrs { global_bytes X, X2, W; };
    auto layer_ptrs = [&](int l) {
        const Layer& L = a.layer[l];
        return LayerPtrs{uniform_global(L.X), uniform_global(L.X2), uniform_global(L.W)};
    };
    struct Ahead {
        global_bytes x;       // piece h' = 0 of sample tile 0 of the wave
        unsigned x_stride;    // bytes from sample tile to sample tile
        unsigned off[2];      // per-lane byte offsets of the two pieces
        global_bytes w;       // the slab's weight image
    };
    // (b, s): block and K slab; s >= 16: the row-major second operand, k = 16 (s - 16) + 8 g + 4 h' .. +3.
    // The four sample tiles of a wave are consecutive and Ppad is a multiple of 128, so they are clamped as one.
    auto ahead_of = [&](const LayerPtrs& P, int b, int s) {
        const long blk = blockIdx.x + (long)(b < my_blocks ? b : my_blocks - 1) * gridDim.x;
        long t0 = blk * 8 + wp_u * 4;
        if constexpr (FLAGS & kNoDupX) t0 += uniform(wn) * 8 * (long)gridDim.x;
        if (t0 >= n_tiles) t0 = n_tiles - 4;
        if constexpr (FLAGS & kSameX) t0 = 0;
        // (selects, not branches: control flow here would cut the MFMA stream into scheduling regions)
        const bool main_part = s < 16;
        int sx = s;
        if constexpr (FLAGS & kRotate) sx = (s + (int)blk * 5) & 15;
        const global_bytes p1 = P.X + t0 * 32768 + sx * 2048;
        const global_bytes p2 = P.X2 + t0 * 128 * a.x2_ld + (s - 16) * 64;
        Ahead A;
        A.x = uniform_global(main_part ? p1 : p2);
        A.x_stride = main_part ? 32768u : 128u * (unsigned)a.x2_ld;
        // (pinned: for the two peeled slabs of a block the select is block-invariant -- hoisted, it spills)
        const unsigned xa = pinned_here(xoff_a), xb = pinned_here(xoff_b);
        A.off[0] = main_part ? xa : xb;
        A.off[1] = A.off[0] + (main_part ? 512u : 16u);
        A.w = uniform_global(P.W + (long)s * (kSlabShorts * 2));
        if constexpr (FLAGS & kXfromW) {
            A.x = A.w + wp_u * 8192;
            A.x_stride = 2048u;
            A.off[0] = (unsigned)lane * 16u;
            A.off[1] = A.off[0] + 1024u;
        }
        return A;
    };
    f32x4 raw[2][4][2];                             // [set = slab parity][sample tile][h']
    auto load_x = [&](auto set_tag, const Ahead& A, auto j_tag, auto half_tag) {
        constexpr int SET = decltype(set_tag)::value, j = decltype(j_tag)::value, HALF = decltype(half_tag)::value;
        if constexpr (!(FLAGS & kNoCut)) {
            // (non-temporal loads: no difference; the two waves that share a sample tile meet in L1 either way)
            raw[SET][j][HALF] = load_f32x4(uniform_global(A.x + j * A.x_stride), A.off[HALF]);
        }
    };
    // planes of the fragment being used / being cut: [ping-pong][plane]
    s16x8 xp[2][3];
    unsigned cu[8], c1[8], c2[8];
    // step 0..7: element e (and / sub / and / sub); 8, 9, 10: pack plane h, m, l (4 v_perm each)
    auto cut_step = [&](auto set_tag, auto j_tag, auto dst_tag, auto step_tag, float x_scale) {
        constexpr int SET = decltype(set_tag)::value, j = decltype(j_tag)::value, D = decltype(dst_tag)::value,
                      STEP = decltype(step_tag)::value;
        if constexpr (FLAGS & kHalf3) {
            if constexpr (STEP < 8) {
                const float x = raw[SET][j][STEP >> 2][STEP & 3] * x_scale;
                cu[STEP] = f16_bits(x);
                c1[STEP] = f16_bits(x - f16_value(cu[STEP]));          // (the difference is exact)
            } else if constexpr (STEP < 10) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const unsigned (&src)[8] = STEP == 8 ? cu : c1;
                const u32x4 v = {low_halves(src[0], src[1]), low_halves(src[2], src[3]),
                                 low_halves(src[4], src[5]), low_halves(src[6], src[7])};
                xp[D][STEP - 8] = __builtin_bit_cast(s16x8, v);
            }
        } else if constexpr (FLAGS & kNoCutMath) {
            if constexpr (STEP == 8) {
                xp[D][0] = __builtin_bit_cast(s16x8, raw[SET][j][0]);
                xp[D][1] = __builtin_bit_cast(s16x8, raw[SET][j][1]);
                xp[D][2] = __builtin_bit_cast(s16x8, raw[SET][j][0]);
            }
        } else if constexpr (!(FLAGS & kNoCut)) {
            if constexpr (STEP < 8) {
                const float x = raw[SET][j][STEP >> 2][STEP & 3];
                cu[STEP] = __float_as_uint(x);
                const float d1 = x - __uint_as_float(cu[STEP] & 0xffff0000u);
                c1[STEP] = __float_as_uint(d1);
                const float d2 = d1 - __uint_as_float(c1[STEP] & 0xffff0000u);
                c2[STEP] = __float_as_uint(d2);
            } else {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                auto pack = [&](const unsigned (&src)[8]) {
                    const u32x4 v = {high_halves(src[0], src[1]), high_halves(src[2], src[3]),
                                     high_halves(src[4], src[5]), high_halves(src[6], src[7])};
                    return __builtin_bit_cast(s16x8, v);
                };
                if constexpr (STEP == 8) xp[D][0] = pack(cu);
                else if constexpr (STEP == 9) xp[D][1] = pack(c1);
                else xp[D][2] = pack(c2);
            }
        }
    };

    // ---- per-sample scales (kHalf3): xs[j] for the block in work, xn[j] for its successor (fetched a block ahead)
    float xs[4] = {1.f, 1.f, 1.f, 1.f}, xn[4] = {1.f, 1.f, 1.f, 1.f};
    // 2^k with |x| 2^k < 2^13 for every |x| <= max(a0, a1): exponent field e of the maximum -> (266 - e) << 23;
    // maxima below 2^-115 (zero, denormals) take the largest scale: nothing to represent there
    auto scale_from = [&](float a0, float a1) {
        const unsigned e = (__float_as_uint(fmaxf(a0, a1)) >> 23) & 0xffu;
        return __uint_as_float((266u - (e < 13u ? 13u : e)) << 23);
    };
    // The maxima of a block that an EARLIER layer of this launch stored (l > 0) are taken from LDS (slot = the block's
    // place in its group; the producing epilogue is at least a block, i.e. many barriers, back): through memory they
    // would sit behind that epilogue's 256 KB of stores with nothing to order them against this read.  Layer 0 of a
    // launch reads what an earlier launch left in memory.  (Groups longer than kAmaxSlots: memory, three blocks later.)
    auto fetch_scales = [&](float (&dst)[4], int l, int b, int slot) {
        if constexpr (FLAGS & kHalf3) {
            const Layer& L = a.layer[l < n_layers ? l : n_layers - 1];
            const long blk = blockIdx.x + (long)(b < my_blocks ? b : my_blocks - 1) * gridDim.x;
            long t0 = blk * 8 + wp_u * 4;
            if (t0 >= n_tiles) t0 = n_tiles - 4;
            if (L.amax_in && l > 0 && slot >= 0) {
                const float* t = lds_amax + slot * 512 + wp_u * 128 + (pinned_here((unsigned)lane) & 31u);
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[j] = scale_from(t[j * 32], t[256 + j * 32]);
            } else if (L.amax_in) {
                const global_bytes base = uniform_global(L.amax_in + t0 * 32);
                const unsigned off = (pinned_here((unsigned)lane) & 31u) * 4u;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dst[j] = scale_from(load_at<float>(base, off + j * 128u),
                                        load_at<float>(base + a.Ppad * 4, off + j * 128u));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[j] = a.x_scale;
            }
        }
    };

    // ---- W: slab image copy (6 x 16 bytes per thread) and fragment reads
    f32x4 wst[6];
    const unsigned woff = (unsigned)tid * 16u;
    auto load_w = [&](const Ahead& A, auto x_tag) {
        constexpr int x = decltype(x_tag)::value;
        wst[x] = load_f32x4(A.w + x * (kThreads * 16), woff);
    };
    auto write_w = [&](int buf, auto x_tag) {
        constexpr int x = decltype(x_tag)::value;
        *(reinterpret_cast<f32x4*>(lds + buf * kSlabShorts) + x * kThreads + tid) = wst[x];
    };
    s16x8 wf[3][4];                                 // [plane][feature tile]
    auto read_w = [&](int buf, auto pl_tag, auto i_tag) {
        constexpr int pl = decltype(pl_tag)::value, i = decltype(i_tag)::value;
        wf[pl][i] = *(reinterpret_cast<const s16x8*>(lds + buf * kSlabShorts) + (pl * 8 + 4 * wn + i) * 64 + lane);
    };
    auto sync = [&]() { if constexpr (!(FLAGS & kNoBarrier)) block_sync(); };

    // one sample tile of one slab: 24 MFMAs; `filler(slot)` goes in front of MFMA `slot`
    auto iteration = [&](auto j_tag, auto pp_tag, auto first_tag, auto filler) {
        constexpr int j = decltype(j_tag)::value, PP = decltype(pp_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        auto one = [&](auto slot_tag) {
            constexpr int S = decltype(slot_tag)::value;
            constexpr int prod = S >> 2, i = S & 3;
            constexpr int wpl = prod < 3 ? 0 : prod < 5 ? 1 : 2;                 // Wh Wh Wh Wm Wm Wl
            constexpr int xpl = prod == 0 ? 0 : prod == 1 ? 1 : prod == 2 ? 2 : prod == 3 ? 0 : prod == 4 ? 1 : 0;
            filler(slot_tag);
            sched_fence();
            if constexpr (FIRST && prod == 0) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[i][j] = mfma_32x32x16_bf16(wf[wpl][i], xp[PP][xpl], zero);
            } else {
                acc[i][j] = mfma_32x32x16_bf16(wf[wpl][i], xp[PP][xpl], acc[i][j]);
            }
            sched_fence();
        };
        if constexpr (FLAGS & kHalf3) {
            auto one3 = [&](auto slot_tag) {
                constexpr int S = decltype(slot_tag)::value;
                constexpr int prod = S >> 2, i = S & 3;
                constexpr int wpl = prod == 2 ? 1 : 0, xpl = prod == 1 ? 1 : 0;          // (Wh Xh)(Wh Xl)(Wl Xh)
                filler(slot_tag);
                sched_fence();
                if constexpr (FIRST && prod == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[i][j] = mfma_32x32x16_f16(wf[wpl][i], xp[PP][xpl], zero);
                } else {
                    acc[i][j] = mfma_32x32x16_f16(wf[wpl][i], xp[PP][xpl], acc[i][j]);
                }
                sched_fence();
            };
            one3(I<0>{}); one3(I<1>{}); one3(I<2>{}); one3(I<3>{}); one3(I<4>{}); one3(I<5>{});
            one3(I<6>{}); one3(I<7>{}); one3(I<8>{}); one3(I<9>{}); one3(I<10>{}); one3(I<11>{});
        } else {
        one(I<0>{}); one(I<1>{}); one(I<2>{}); one(I<3>{}); one(I<4>{}); one(I<5>{}); one(I<6>{}); one(I<7>{});
        one(I<8>{}); one(I<9>{}); one(I<10>{}); one(I<11>{}); one(I<12>{}); one(I<13>{}); one(I<14>{}); one(I<15>{});
        one(I<16>{}); one(I<17>{}); one(I<18>{}); one(I<19>{}); one(I<20>{}); one(I<21>{}); one(I<22>{}); one(I<23>{});
        }
    };

    // slab u out of LDS buffer BUF (= u & 1 = raw set): per sample tile j
    //   slots 0-10   cut the next fragment ((u, j + 1), or (u + 1, 0) out of the other raw set)
    //   slots 6, 18  reload the two pieces of raw fragment j for the slab two ahead
    //   slots 0, 12  j = 0, 1, 2: write piece 2 j, 2 j + 1 of W(u + 1) (loaded during slab u - 1) into buffer BUF ^ 1;
    //                j = 1, 2, 3: load piece 2 (j - 1), 2 (j - 1) + 1 of W(u + 2) into the staging register just written
    //   j = 0: slots 2, 6, 10, 14 read Wl(u);   j = 3: slots 12, 14, 16, 18 read Wh(u + 1), slots 20-23 Wm(u + 1)
    //   j = 2: barrier at the end (every wave is done with W(u): its buffer is free for slab u + 1's writes; W(u + 1),
    //          written during this slab, is visible)
    // One memory instruction per wave every 6 slots: the four waves issue theirs in lock-step, 4 KB per slot position
    // = 64 clocks of the CU's 64 B/clk address unit = two MFMA slots; issued back to back (six weight pieces in six
    // slots, two activation pieces in one) the address unit's queue filled and the waves stalled in front of their
    // next MFMA -- the unit's busy time was exposed nearly one to one (lab: weights +0.054 ms, activations even when
    // they always hit in L1 +0.047 ms per layer on top of 0.290 for the MFMAs alone).
    // vmcnt counts loads IN ORDER: waiting for a weight piece also waits for every activation load issued before it;
    // a piece is written 72 slots (~1.1 us) after its load.
    auto slab = [&](auto buf_tag, auto first_tag, const LayerPtrs& cur, const LayerPtrs& nxt, bool last_layer, int n_k,
                    int b0, int b1, int b, int s) {
        constexpr int BUF = decltype(buf_tag)::value;
        using Set = I<BUF>;
        using Other = I<BUF ^ 1>;
        // two slabs ahead: this block, or slab 0 / 1 of the group's next block, or of its first block in the next
        // layer (nxt), or of the next group's first block in layer 0 (nxt again; past the end: clamped, never used)
        const bool wraps = s + 2 >= n_k;
        const bool last_of_group = b + 1 >= b1;
        const bool to_next = wraps && last_of_group;
        const int b2 = !wraps ? b : !last_of_group ? b + 1 : last_layer ? b1 : b0;
        const int s2 = wraps ? s + 2 - n_k : s + 2;
        const LayerPtrs P{to_next ? nxt.X : cur.X, to_next ? nxt.X2 : cur.X2, to_next ? nxt.W : cur.W};
        const Ahead A = ahead_of(P, b2, s2);
        auto fill = [&](auto j_tag) {
            return [&](auto slot_tag) {
                constexpr int j = decltype(j_tag)::value, S = decltype(slot_tag)::value;
                if constexpr (FLAGS & kHalf3) {         // 12 slots per sample tile: the same fillers, twice as dense
                    if constexpr (S <= 9) {
                        // (the next slab's first fragment: in the block's last slab it belongs to the successor block)
                        if constexpr (j < 3) cut_step(Set{}, I<j + 1>{}, I<(j + 1) & 1>{}, I<S>{}, xs[j + 1]);
                        else cut_step(Other{}, I<0>{}, I<0>{}, I<S>{}, s + 1 >= n_k ? xn[0] : xs[0]);
                    }
                    if constexpr (S == 3) load_x(Set{}, A, I<j>{}, I<0>{});
                    if constexpr (S == 9) load_x(Set{}, A, I<j>{}, I<1>{});
                    if constexpr ((S == 0 || S == 6) && !(FLAGS & kNoWCopy)) {
                        if constexpr (j <= 2) write_w(BUF ^ 1, I<2 * j + (S == 6)>{});
                        if constexpr (j >= 1) load_w(A, I<2 * (j - 1) + (S == 6)>{});
                    }
                    if constexpr (j == 0 && S >= 1 && S <= 4) read_w(BUF, I<1>{}, I<S - 1>{});
                    if constexpr (j == 3 && S >= 8) read_w(BUF ^ 1, I<0>{}, I<S - 8>{});
                } else {
                if constexpr (S <= 10) {
                    if constexpr (j < 3) cut_step(Set{}, I<j + 1>{}, I<(j + 1) & 1>{}, I<S>{}, 1.f);
                    else cut_step(Other{}, I<0>{}, I<0>{}, I<S>{}, 1.f);
                }
                if constexpr (S == 6) load_x(Set{}, A, I<j>{}, I<0>{});
                if constexpr (S == 18) load_x(Set{}, A, I<j>{}, I<1>{});
                if constexpr ((S == 0 || S == 12) && !(FLAGS & kNoWCopy)) {
                    if constexpr (j <= 2) write_w(BUF ^ 1, I<2 * j + (S == 12)>{});
                    if constexpr (j >= 1) load_w(A, I<2 * (j - 1) + (S == 12)>{});
                }
                if constexpr (j == 0) {
                    if constexpr (S == 2 || S == 6 || S == 10 || S == 14) read_w(BUF, I<2>{}, I<(S - 2) / 4>{});
                } else if constexpr (j == 3) {
                    if constexpr (S == 12 || S == 14 || S == 16 || S == 18) read_w(BUF ^ 1, I<0>{}, I<(S - 12) / 2>{});
                    if constexpr (S >= 20) read_w(BUF ^ 1, I<1>{}, I<S - 20>{});
                }
                }
            };
        };
        iteration(I<0>{}, I<0>{}, first_tag, fill(I<0>{}));
        iteration(I<1>{}, I<1>{}, first_tag, fill(I<1>{}));
        iteration(I<2>{}, I<0>{}, first_tag, fill(I<2>{}));
        sync();
        iteration(I<3>{}, I<1>{}, first_tag, fill(I<3>{}));
    };

    // finished block: bias, activation, ReLU bits, tile-native stores.  Piece (i, q) -- 16 bytes per lane, features
    // 32 (4 wn + i) + 8 q + 4 g .. +3 -- of all FOUR sample tiles at a time: its bias piece is read from the LDS table
    // once (and one piece ahead: read right where it is used, every piece waited out an LDS round trip -- 64 of them
    // per block and wave were nearly half of the epilogue's 14.7 k cycles).  From LDS, not from memory: a global load
    // here waits on vmcnt, which counts the stores just issued too.  The stores go through a wave-uniform base + a
    // 32-bit lane offset recomputed here: as 64-bit per-lane pointers their lane-invariant parts were hoisted out of
    // the block loop and spilled, and a scratch reload between the stores waits for all of them.  The fence keeps the
    // accumulator reads piece by piece: hoisted, they spill.
    auto epilogue = [&](int l, int b, int slot) {
        const Layer& L = a.layer[l];
        const long blk = blockIdx.x + (long)b * gridDim.x;
        const long tile0 = blk * 8 + wp_u * 4;      // Ppad is a multiple of 128: a wave's four tiles are in range together
        if (tile0 >= n_tiles) return;               // (wave-uniform: a scalar branch)
        const int wn_u = uniform(wn);
        const global_bytes_rw z0 = uniform_global_rw(L.Z + tile0 * 8192 + wn_u * 4096);     // + j * 32 KB + (4 i + q) * 1 KB
        const unsigned lane_here = pinned_here((unsigned)lane);       // (lane-invariant values derived outside get hoisted
        const unsigned zoff = lane_here * 16u;                        //  out of the block loop into registers that spill)
        const float* const table = lds_bias + (16 * wn_u) * 8 + (lane_here >> 5) * 4;       // + (4 i + q) * 8
        auto pieces = [&](auto f) {
            f(I<0>{}); f(I<1>{}); f(I<2>{}); f(I<3>{}); f(I<4>{}); f(I<5>{}); f(I<6>{}); f(I<7>{});
            f(I<8>{}); f(I<9>{}); f(I<10>{}); f(I<11>{}); f(I<12>{}); f(I<13>{}); f(I<14>{}); f(I<15>{});
        };
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        // kHalf3: what the accumulators are multiplied by, per sample: 1 / (sample scale x weight scale), powers of two
        float os[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (FLAGS & kHalf3) {
            const float w_inv = L.w_inv_scale ? *L.w_inv_scale : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                os[j] = L.w_inv_scale ? __uint_as_float(0x7f000000u - __float_as_uint(xs[j])) * w_inv : a.out_scale;
        }
        // kAmaxOut: max |stored value| per sample over this wave's 128 features -> amax_out[wn][sample]
        auto store_amax = [&](int j, float hm) {
            if constexpr (FLAGS & kAmaxOut) {
                if (L.amax_out) {
                    const float both = fmaxf(hm, shfl_xor(hm, 32));
                    if (lane_here < 32u) {
                        store_at(uniform_global_rw(L.amax_out + wn_u * a.Ppad + (tile0 + j) * 32), lane_here * 4u, both);
                        if (slot >= 0) lds_amax[slot * 512 + wn_u * 256 + wp_u * 128 + j * 32 + lane_here] = both;
                    }
                }
            }
        };
        if (a.mode == 0) {
            const float lo = L.relu ? 0.f : -__builtin_huge_valf();
            auto tile_pair = [&](auto j0_tag) {
            constexpr int J0 = decltype(j0_tag)::value;
            unsigned hb[4] = {0u, 0u, 0u, 0u}, bits[4][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};
            float hm[4] = {0.f, 0.f, 0.f, 0.f};
            f32x4 b_next = *reinterpret_cast<const f32x4*>(table);
            pieces([&](auto idx_tag) {
                constexpr int IDX = decltype(idx_tag)::value, i = IDX >> 2, q = IDX & 3;
                sched_fence();
                const f32x4 bc = b_next;
                if constexpr (IDX < 15) b_next = *reinterpret_cast<const f32x4*>(table + (IDX + 1) * 8);
#pragma unroll
                for (int j = J0; j < J0 + 2; ++j) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (FLAGS & kHalf3) v[e] = max_raw(fmaf(acc[i][j][4 * q + e], os[j], bc[e]), lo);
                        else v[e] = max_raw(add_raw(acc[i][j][4 * q + e], bc[e]), lo);
                        hb[j] = shift_in_positive(hb[j], v[e]);
                        if constexpr (FLAGS & kAmaxOut) hm[j] = fmaxf(hm[j], fabsf(v[e]));
                    }
                    const unsigned dst = zoff + (unsigned)IDX * 1024u;
                    if constexpr (FLAGS & kNoZStore) { (void)dst; }
                    else if constexpr (FLAGS & kPlainStore) store_at(z0 + j * 32768, dst, v);
                    else store_stream_at(z0 + j * 32768, dst, v);
                }
                if constexpr (q == 3) {
#pragma unroll
                    for (int j = J0; j < J0 + 2; ++j) {
                        bits[j][i >> 1] = (bits[j][i >> 1] << 16) | hb[j];
                        hb[j] = 0u;
                    }
                }
            });
            if (L.mask) {
#pragma unroll
                for (int j = J0; j < J0 + 2; ++j)
                    store_at(uniform_global_rw(L.mask + (tile0 + j) * 256 + 2 * wn_u), zoff, u32x2{bits[j][0], bits[j][1]});
            }
#pragma unroll
            for (int j = J0; j < J0 + 2; ++j) store_amax(j, hm[j]);
            };
            tile_pair(I<0>{});       // (two tiles at a time: all four hold 16 accumulator values + their results per
            tile_pair(I<2>{});       //  piece next to the loop's prefetched operands -- the allocator spills 32 registers)
            return;
        }
        // data-gradient form: gate by the ReLU bits of the layer below (element 16 t + r of a lane: word t >> 1, bit
        // 31 - (16 (t & 1) + r)), after adding the density head's rank-1 term.  The gates and the per-sample scalars of
        // the four tiles are fetched up front (one wait, before this block's first store).
        auto tile_pair_bwd = [&](auto j0_tag) {
        constexpr int J0 = decltype(j0_tag)::value;
        u32x2 gate[4];
        float vs[4];
#pragma unroll
        for (int j = J0; j < J0 + 2; ++j) {
            gate[j] = load_at<u32x2>(uniform_global(L.mask_in + (tile0 + j) * 256 + 2 * wn_u), zoff);
            const unsigned p_in = (unsigned)j * 32u + (lane_here & 31u);          // sample within the wave's four tiles
            vs[j] = (L.vec && tile0 * 32 + p_in < a.n_vec)
                        ? load_at<float>(uniform_global(L.vec + tile0 * 32 * a.vec_stride), p_in * (unsigned)a.vec_stride * 4u)
                        : 0.f;
        }
        // RANK1: the density head's term (feature_linear^T only); the other seven layers skip the FMA and the table
        float hm[4] = {0.f, 0.f, 0.f, 0.f};
        auto all_pieces = [&](auto rank1_tag) {
            constexpr bool RANK1 = decltype(rank1_tag)::value;
            f32x4 b_next = {0.f, 0.f, 0.f, 0.f};
            if constexpr (RANK1) b_next = *reinterpret_cast<const f32x4*>(table);
            pieces([&](auto idx_tag) {
                constexpr int IDX = decltype(idx_tag)::value, i = IDX >> 2, q = IDX & 3;
                sched_fence();
                const f32x4 bc = b_next;
                if constexpr (RANK1 && IDX < 15) b_next = *reinterpret_cast<const f32x4*>(table + (IDX + 1) * 8);
                constexpr int bit0 = 31 - (16 * (i & 1) + 4 * q);
#pragma unroll
                for (int j = J0; j < J0 + 2; ++j) {
                    const unsigned word = gate[j][i >> 1];
                    auto val = [&](int e) {
                        const float d = (FLAGS & kHalf3) ? acc[i][j][4 * q + e] * os[j] : acc[i][j][4 * q + e];
                        return RANK1 ? fmaf(bc[e], vs[j], d) : d;
                    };
                    f32x4 v;
                    v[0] = keep_if_bit<bit0 - 0>(val(0), word);
                    v[1] = keep_if_bit<bit0 - 1>(val(1), word);
                    v[2] = keep_if_bit<bit0 - 2>(val(2), word);
                    v[3] = keep_if_bit<bit0 - 3>(val(3), word);
                    if constexpr (FLAGS & kAmaxOut)
                        hm[j] = fmaxf(fmaxf(hm[j], fabsf(v[0])), fmaxf(fmaxf(fabsf(v[1]), fabsf(v[2])), fabsf(v[3])));
                    if constexpr (FLAGS & kPlainStore) store_at(z0 + j * 32768, zoff + (unsigned)IDX * 1024u, v);
                    else store_stream_at(z0 + j * 32768, zoff + (unsigned)IDX * 1024u, v);
                }
            });
        };
        if (L.vec) all_pieces(std::true_type{}); else all_pieces(std::false_type{});
#pragma unroll
        for (int j = J0; j < J0 + 2; ++j) store_amax(j, hm[j]);
        };
        tile_pair_bwd(I<0>{});
        tile_pair_bwd(I<2>{});
    };

    unsigned long long probe_c = 0, probe_r = 0;
    if constexpr (FLAGS & kClockProbe) { probe_c = shader_cycles(); probe_r = reference_ticks(); }
    lds_bias[tid] = a.layer[0].bias[tid];
    // ---- prologue: W slabs 0 and 1 -> buffers 0 and 1; X raw of slabs 0 and 1; first fragment cut
    {
        const LayerPtrs P0 = layer_ptrs(0);
        const Ahead A0 = ahead_of(P0, 0, 0), A1 = ahead_of(P0, 0, 1);
        load_w(A0, I<0>{}); load_w(A0, I<1>{}); load_w(A0, I<2>{}); load_w(A0, I<3>{}); load_w(A0, I<4>{}); load_w(A0, I<5>{});
        load_x(I<0>{}, A0, I<0>{}, I<0>{}); load_x(I<0>{}, A0, I<1>{}, I<0>{}); load_x(I<0>{}, A0, I<2>{}, I<0>{}); load_x(I<0>{}, A0, I<3>{}, I<0>{});
        load_x(I<0>{}, A0, I<0>{}, I<1>{}); load_x(I<0>{}, A0, I<1>{}, I<1>{}); load_x(I<0>{}, A0, I<2>{}, I<1>{}); load_x(I<0>{}, A0, I<3>{}, I<1>{});
        load_x(I<1>{}, A1, I<0>{}, I<0>{}); load_x(I<1>{}, A1, I<1>{}, I<0>{}); load_x(I<1>{}, A1, I<2>{}, I<0>{}); load_x(I<1>{}, A1, I<3>{}, I<0>{});
        load_x(I<1>{}, A1, I<0>{}, I<1>{}); load_x(I<1>{}, A1, I<1>{}, I<1>{}); load_x(I<1>{}, A1, I<2>{}, I<1>{}); load_x(I<1>{}, A1, I<3>{}, I<1>{});
        write_w(0, I<0>{}); write_w(0, I<1>{}); write_w(0, I<2>{}); write_w(0, I<3>{}); write_w(0, I<4>{}); write_w(0, I<5>{});
        // (slab 1's image stays in the staging registers: slab 0 writes it, as every slab writes its successor's)
        load_w(A1, I<0>{}); load_w(A1, I<1>{}); load_w(A1, I<2>{}); load_w(A1, I<3>{}); load_w(A1, I<4>{}); load_w(A1, I<5>{});
    }
    sync();
    read_w(0, I<0>{}, I<0>{}); read_w(0, I<0>{}, I<1>{}); read_w(0, I<0>{}, I<2>{}); read_w(0, I<0>{}, I<3>{});
    read_w(0, I<1>{}, I<0>{}); read_w(0, I<1>{}, I<1>{}); read_w(0, I<1>{}, I<2>{}); read_w(0, I<1>{}, I<3>{});
    {
        fetch_scales(xs, 0, 0, -1);
        auto whole = [&](auto s) { cut_step(I<0>{}, I<0>{}, I<0>{}, s, xs[0]); };
        whole(I<0>{}); whole(I<1>{}); whole(I<2>{}); whole(I<3>{}); whole(I<4>{}); whole(I<5>{}); whole(I<6>{});
        whole(I<7>{}); whole(I<8>{}); whole(I<9>{}); whole(I<10>{});
    }

    bool first_table = true;
    if constexpr (FLAGS & kRandomX) {
        unsigned h = (unsigned)tid * 2654435761u + 12345u;
        for (int d = 0; d < 2; ++d)
            for (int pl = 0; pl < 3; ++pl)
                for (int e = 0; e < 8; ++e) {
                    h = h * 1664525u + 1013904223u;
                    xp[d][pl][e] = (short)((h >> 16) & 0xbfff);        // (exponent below 2: no overflow over the K loop)
                }
    }
    for (int b0 = 0, b1; b0 < my_blocks; b0 = b1) {
        b1 = b0 + a.group;
        if (b1 + 2 > my_blocks) b1 = my_blocks;             // (a stray last block joins the last group)
        for (int l = 0; l < n_layers; ++l) {
            const int n_k = a.layer[l].n_k;
            const bool last_layer = l + 1 >= n_layers;
            const LayerPtrs cur = layer_ptrs(l), nxt = layer_ptrs(last_layer ? 0 : l + 1);
            if (!first_table) {
                sync();                                     // every wave is done with the previous layer's bias table
                lds_bias[tid] = a.layer[l].bias[tid];       // (visible long before the first epilogue: a barrier per slab)
            }
            first_table = false;
            for (int b = b0; b < b1; ++b) {
                slab(I<0>{}, std::true_type{}, cur, nxt, last_layer, n_k, b0, b1, b, 0);
                // (kHalf3) the successor block's scales, most of a block ahead of their first use -- and behind slab 0's
                // barrier: the epilogue that has just run may be the one that left them in LDS
                const bool slots = b1 - b0 <= kAmaxSlots;
                fetch_scales(xn, b + 1 < b1 ? l : last_layer ? 0 : l + 1, b + 1 < b1 ? b + 1 : last_layer ? b1 : b0,
                             !slots ? -1 : b + 1 < b1 ? b + 1 - b0 : 0);
                slab(I<1>{}, std::false_type{}, cur, nxt, last_layer, n_k, b0, b1, b, 1);
                for (int s = 2; s < n_k; s += 2) {
                    slab(I<0>{}, std::false_type{}, cur, nxt, last_layer, n_k, b0, b1, b, s);
                    slab(I<1>{}, std::false_type{}, cur, nxt, last_layer, n_k, b0, b1, b, s + 1);
                }
                if constexpr (!(FLAGS & kNoEpilogue)) epilogue(l, b, b1 - b0 <= kAmaxSlots ? b - b0 : -1);
                if constexpr (FLAGS & kHalf3) { xs[0] = xn[0]; xs[1] = xn[1]; xs[2] = xn[2]; xs[3] = xn[3]; }
            }
        }
    }
    if constexpr (FLAGS & kNoEpilogue) epilogue(n_layers - 1, my_blocks - 1, -1);
    if constexpr (FLAGS & kClockProbe) {
        if (tid == 0) {
            a.clock_probe[2 * blockIdx.x] = shader_cycles() - probe_c;
            a.clock_probe[2 * blockIdx.x + 1] = reference_ticks() - probe_r;
        }
    }       // (keeps the MFMAs alive)
}

}  // namespace lsp
}  // namespace scn
