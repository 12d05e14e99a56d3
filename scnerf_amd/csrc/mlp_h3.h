// mlp_h3.h -- the machinery of the "resident" arithmetic: the whole network in ONE launch on three fp16 products per
// product (v_mfma_f32_32x32x16_f16), with the activations (forward) / output gradients (data-gradient chain) of a
// wave's 32 samples REGISTER-RESIDENT from layer to layer as already-cut fp16 planes.  Used by mlp_fwd_h3.hip and
// mlp_bwd_h3.hip; the weight stream's format is described in scnerf_amd/mlp_layout.py (h3_plan).
//
// Why.  layer_split.h runs a 256 -> 256 layer as a GEMM over all samples: every layer's activations travel to HBM
// and back (the step moved ~70 GB, 3.5 TB/s: HBM-bound), the consumer cuts them again in two waves, and the fused
// kernels' end stages re-read what the last GEMM had in its accumulators.  Here a layer's result never leaves the
// register file on its way to the next layer: it is WRITTEN once for the weight-gradient GEMMs (training) and that is
// all the activation traffic there is.
//
// Arithmetic (as layer_split.h's kHalf3: per-GEMM error against fp64 equal to the fp32 MFMA's): every operand is
// scaled by a power of two and cut into two fp16 numbers, x S = h + l (|l| <= 2^-11 |h|, the cut exact to 2^-22);
// a product is (Wh Xh) + (Wh Xl) + (Wl Xh) accumulated in fp32 -- the dropped (Wl Xl) is 2^-22 of the product.
// Scales: one per layer for the weights (|w| Sw < 2^13, packing pass), one per SAMPLE for the register-resident
// operand -- a sample is a column of the transposed product, so its scale comes back out lane-wise in the epilogue.
//
// The per-sample scale without waiting for the layer's own maximum.  The cut of a layer's output needs a power of
// two S with |z| S < 2^13 for all 256 features of the sample -- known only when the whole layer is done, which would
// serialise epilogue and next layer.  Instead:   |z_n| = |sum_k w_nk x_k + b_n| <= A max_k|x_k| + B,
//     A = max_n sum_k |w_nk|  (largest row 1-norm),  B = max_n |b_n|   (scale pass, once per optimizer step)
// with max|x| the MEASURED maximum of the layer's INPUT (one v_max3 per two elements in the producing epilogue).
// The bound is known before the layer's first MFMA and is loose by one layer only (typically 4-16x); fp16 leaves
// room: a value cut at scale S carries an absolute error <= max(2^-22 |x S|, 2^-25), so the cut stays at fp32 grade
// while the sample's true maximum times S is anywhere in [2^-3, 2^13] -- ten octaves below the bound.
// (Data gradients: the same with the column 1-norms A' of the layer.)
//
// Shape.  Wave = 32 samples x all features, one wave per SIMD, 4 waves per workgroup sharing the weight stream.
// The product is computed transposed, D[feature][sample]: lane (m, h) owns feature 32 t + 8 q + 4 h + j of sample m in
// register 4 q + j of output tile t (mlp_common.h).  The fp16 MFMA contracts 16 k per instruction, 8 per lane half:
// the two 16-byte pieces (t, 2 u), (t, 2 u + 1) of a lane ARE its 8 elements of K slab 2 t + u once the weights are
// packed in that order -- the cut epilogue writes B operands directly.
// Output tiles are produced PAIR by PAIR (all K slabs of tiles 2 P, 2 P + 1, then the next pair), so a pair is final
// after a quarter of the layer and its epilogue -- bias, ReLU, mask bits, maximum, store, cut: ~8.5 VALU per element --
// is issued in slices between the MFMAs of the following pair (a 32-cycle fp16 MFMA hides ~5 other instructions of
// a lone wave).  The LAST pair's epilogue runs under the first pair of the NEXT layer, whose K loop reaches the
// slabs that epilogue produces (tiles 6, 7 = slabs 12 .. 15) only after 72 MFMAs.  Two operand buffers (256
// registers) + two accumulator pairs (64) + fragment ring (32) + stream staging (32): ~430 of 512 registers.
//
// Weight stream.  16-byte A fragments in consumption order, 32 KB chunks through three LDS buffers (mlp_common.h's
// discipline: chunk c + 1 is written while chunk c is consumed, one barrier per chunk at slot 30 of 48); a chunk is
// only 48 MFMAs long, so its global loads are issued a whole chunk before their LDS writes (slots 24 .. 47 of chunk
// c - 1 -> slots 0 .. 23 of chunk c), one memory instruction every third slot.
#pragma once
#include <type_traits>
#include <utility>

#include <scn_lab.h>
#include <scn_wave.h>

#include "mlp_common.h"

namespace scn {
namespace h3 {

using namespace scn::mlp;

constexpr int kChunkBytes = 32768;
constexpr int kStreamLds = 3 * kChunkBytes;
constexpr int kTableFloats = 2816;                 // the lane-vector tables of the packed buffer (2724 floats), padded
constexpr int kScaleStride = 8;
enum : int { kSw = 0, kSwInv = 1, kBoundA = 2, kBoundB = 3, kBoundAT = 4, kPoison = 7 };   // (kPoison: the rgb layer's row only)
enum : int { kLayerFeat = 8, kLayerViews = 9, kLayerRgb = 10, kLayerAlpha = 11, kScaleLayers = 12 };

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int N> using I = std::integral_constant<int, N>;

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(I<Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// 2^k with bound 2^k < 2^13 (exponent field e of the bound -> (266 - e) << 23); bounds below 2^-114 take 2^126
__device__ __forceinline__ float scale_for(float bound) {
    const unsigned e = (__float_as_uint(bound) >> 23) & 0xffu;
    return __uint_as_float((266u - (e < 13u ? 13u : e)) << 23);
}
__device__ __forceinline__ float inv_pow2(float s) { return __uint_as_float(0x7f000000u - __float_as_uint(s)); }

// x[0 .. 7] s -> the two planes of one K slab's lane operand
__device__ __forceinline__ void cut8(const float* x, float s, u32x4& h, u32x4& l) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned hp = pack_f16_scaled(x[2 * c], x[2 * c + 1], s);
        h[c] = hp;
        l[c] = pack_f16(residual_f16<0>(x[2 * c], s, hp), residual_f16<1>(x[2 * c + 1], s, hp));
    }
}

// ---- weight stream ------------------------------------------------------------------------------------------------
struct Stream {
    global_bytes g;        // the chunk to fetch next (two ahead of the one in use)
    unsigned cur;          // LDS byte offset of the buffer in use: 0, 32 K, 64 K
    f32x4 stage[8];
    __device__ __forceinline__ unsigned next() const { return cur == 2u * kChunkBytes ? 0u : cur + kChunkBytes; }
};

// first chunk -> buffer 0, second chunk -> staging registers (slots 0 .. 23 of chunk 0 commit it); the caller
// synchronises the workgroup before the first fragment read
__device__ __forceinline__ void stream_prime(Stream& ws, const void* stream, char* lds, unsigned tid16) {
    ws.g = uniform_global(stream);
#pragma unroll
    for (int i = 0; i < 8; ++i) ws.stage[i] = load_f32x4(ws.g + i * 4096, tid16);
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(lds + i * 4096 + tid16) = ws.stage[i];
    ws.g = uniform_global(ws.g + kChunkBytes);
#pragma unroll
    for (int i = 0; i < 8; ++i) ws.stage[i] = load_f32x4(ws.g + i * 4096, tid16);
    ws.g = uniform_global(ws.g + kChunkBytes);
    ws.cur = 0u;
    if constexpr (lab::kNoStream) {        // (timing experiments: the stream is off, every buffer holds real fragments)
#pragma unroll
        for (int b = 1; b < 3; ++b)
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(lds + b * kChunkBytes + i * 4096 + tid16) = ws.stage[i];
    }
}

// what the stream does in slot KAPPA (0 .. 47) of a chunk: piece i of the NEXT chunk goes from its staging register to
// LDS in slot 3 i, and the register is refilled right away (slot 3 i + 1) with piece i of the chunk after that -- a load
// is waited for 47 slots (~1.6 us) after it was issued.  vmcnt counts loads AND stores in issue order: the wait also
// covers every activation store issued before the load, and those take an HBM write's time to retire (with the loads
// issued 24 slots ahead the waits cost ~0.4 ms of a 3 ms launch: lab::kLateLoads is that schedule).
template <int KAPPA>
__device__ __forceinline__ void stream_slot(Stream& ws, char* lds, unsigned tid16) {
    if constexpr (!lab::kNoStream && KAPPA < 24 && KAPPA % 3 == 0)
        *reinterpret_cast<f32x4*>(lds + ws.next() + (KAPPA / 3) * 4096 + tid16) = ws.stage[KAPPA / 3];
    if constexpr (!lab::kNoBarrier && KAPPA == 30) block_sync();
    constexpr int LOAD_PIECE = lab::kNoStream ? -1
                               : lab::kLateLoads ? ((KAPPA >= 24 && KAPPA % 3 == 0) ? (KAPPA - 24) / 3 : -1)
                                                 : ((KAPPA < 24 && KAPPA % 3 == 1) ? KAPPA / 3 : -1);
    if constexpr (LOAD_PIECE >= 0)
        // (opaque wave-uniform base + the thread's 32-bit offset: `global_load v, v_off, s[base]`; left to itself the
        //  compiler keeps a 64-bit per-lane pointer across the layer loop and spills it)
        ws.stage[LOAD_PIECE] = load_f32x4(uniform_global(ws.g + LOAD_PIECE * 4096), pinned_here(tid16));
    if constexpr (KAPPA == 47) {
        ws.cur = ws.next();
        ws.g = uniform_global(ws.g + kChunkBytes);
    }
}

__device__ __forceinline__ s16x8 as_frag(u32x4 v) { return __builtin_bit_cast(s16x8, v); }

struct NoFill {
    template <int S> __device__ __forceinline__ void operator()(I<S>) const {}
};

// The per-wave state every part works on.
struct Wave {
    char* lds;
    unsigned tid16, lane16;
    Stream ws;
    s16x8 ring[2][4];      // A fragments of the unit in use / of the next unit
};

// the first unit's fragments (after stream_prime + barrier)
__device__ __forceinline__ void ring_prime(Wave& w) {
#pragma unroll
    for (int j = 0; j < 4; ++j) w.ring[0][j] = *reinterpret_cast<const s16x8*>(w.lds + j * 1024 + w.lane16);
}

// One UNIT of the stream: four fragments, six MFMAs.  U = the unit's index in the whole stream (compile time: its
// place in the chunk is U % 8, its half of the ring U & 1).
//   PAIR:   one K slab of an output-tile pair -- fragments [Wh0 Wh1 Wl0 Wl1]:
//           (Wh0 Xh)(Wh1 Xh)(Wh0 Xl)(Wh1 Xl)(Wl0 Xh)(Wl1 Xh), tile 0 into acc[0], tile 1 into acc[1]
//   !PAIR:  two K slabs A, B of ONE output tile -- fragments [WhA WlA WhB WlB]:
//           (WhA XhA)(WhB XhB)(WhA XlA)(WhB XlB)(WlA XhA)(WlB XhB), slab A into acc[0], slab B into acc[1] (the caller
//           adds the two: consecutive MFMAs never target the same accumulator)
// `fill(I<j>)` is issued in front of MFMA j; slots 0 .. 3 also fetch the NEXT unit's fragments.
template <int U, bool PAIR, bool FIRST, class Fill>
__device__ __forceinline__ void unit(Wave& w, u32x4 xhA, u32x4 xlA, u32x4 xhB, u32x4 xlB, f32x16 (&acc)[2], Fill&& fill) {
    constexpr int PH = U & 7, RP = U & 1;
    static_for<6>([&](auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        // (the stream first: its barrier drains the LDS queue -- `s_waitcnt lgkmcnt(0)` -- so it sits in front of the
        //  reads this slot issues, not behind them)
        stream_slot<PH * 6 + j>(w.ws, w.lds, w.tid16);
        fill(j_tag);
        if constexpr (j < 4) {
            // (the next unit may sit in the next chunk: its buffer is complete behind this chunk's barrier at slot 30.
            //  At j = 5 of unit 7 stream_slot has already switched buffers, hence the reads stay in slots 0 .. 3.)
            constexpr int NPH = (PH + 1) & 7;
            const unsigned buf = PH == 7 ? w.ws.next() : w.ws.cur;
            w.ring[RP ^ 1][j] = *reinterpret_cast<const s16x8*>(w.lds + buf + (NPH * 4 + j) * 1024 + w.lane16);
        }
        sched_fence();
        constexpr int x = j & 1;
        constexpr int fr = PAIR ? (j < 4 ? (j & 1) : 2 + (j & 1)) : (j < 4 ? 2 * (j & 1) : 1 + 2 * (j & 1));
        const u32x4 b = PAIR ? ((j == 2 || j == 3) ? xlA : xhA)
                             : ((j == 2 || j == 3) ? (x ? xlB : xlA) : (x ? xhB : xhA));
        if constexpr (FIRST && j < 2) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[x] = mfma_32x32x16_f16(w.ring[RP][fr], as_frag(b), zero);
        } else {
            acc[x] = mfma_32x32x16_f16(w.ring[RP][fr], as_frag(b), acc[x]);
        }
        sched_fence();
    });
}

// All K slabs of one output-tile pair: NK units starting at stream unit U0.  operand(I<s>, &xh, &xl) names the
// planes of K slab s; fill(I<sigma>) is the filler of slot sigma = 6 s + j.
template <int U0, int NK, class Operand, class Fill>
__device__ __forceinline__ void tile_pair(Wave& w, f32x16 (&acc)[2], Operand&& operand, Fill&& fill) {
    static_for<NK>([&](auto s_tag) {
        constexpr int s = decltype(s_tag)::value;
        u32x4 xh, xl;
        operand(s_tag, xh, xl);
        unit<U0 + s, true, s == 0>(w, xh, xl, xh, xl, acc, [&](auto j_tag) { fill(I<6 * s + decltype(j_tag)::value>{}); });
    });
}

// All K slabs (2 NU of them) of a single output tile; the result is acc[0] + acc[1].
template <int U0, int NU, class Operand, class Fill>
__device__ __forceinline__ void tile_single(Wave& w, f32x16 (&acc)[2], Operand&& operand, Fill&& fill) {
    static_for<NU>([&](auto u_tag) {
        constexpr int u = decltype(u_tag)::value;
        u32x4 xhA, xlA, xhB, xlB;
        operand(I<2 * u>{}, xhA, xlA);
        operand(I<2 * u + 1>{}, xhB, xlB);
        unit<U0 + u, false, u == 0>(w, xhA, xlA, xhB, xlB, acc, [&](auto j_tag) { fill(I<6 * u + decltype(j_tag)::value>{}); });
    });
}

// ---- epilogue slices ------------------------------------------------------------------------------------------------
// An epilogue works on one output-tile pair (tiles 2 P, 2 P + 1 of its layer) in eight PIECES -- (tile x, quarter q):
// four accumulator registers = one 16-byte piece of the tile-native section = half of a K slab's lane operand -- and a
// piece in twelve SUB-steps of at most four VALU instructions.  slot<P, SIGMA, SPP> maps the filler slot SIGMA of the
// covering tile pair to sub-steps: SPP = 12 slots per piece (96 slots: a whole pair of a 16-slab layer) or 9 (72
// slots: the pair that must be done before the next layer's K loop reaches slab 12).
template <class Epi, int P, int SIGMA, int SPP, int NS>
__device__ __forceinline__ void epi_slot(Epi& e, f32x16 (&acc)[2], u32x4 (&oh)[NS], u32x4 (&ol)[NS]) {
    static_assert(SPP == 12 || SPP == 9 || SPP == 6, "slots per piece");
    if constexpr (SIGMA < 8 * SPP) {
        constexpr int piece = SIGMA / SPP, k = SIGMA % SPP;
        if constexpr (SPP == 12) {
            e.template sub<P, piece, k>(acc, oh, ol);
        } else if constexpr (SPP == 6) {       // (a pair of an 8-slab part: 48 slots, two sub-steps each)
            e.template sub<P, piece, 2 * k>(acc, oh, ol);
            e.template sub<P, piece, 2 * k + 1>(acc, oh, ol);
        } else {
            // 12 sub-steps over 9 slots: (0 1)(2)(3)(4)(5 6)(7)(8)(9)(10 11)
            constexpr int first = k == 0 ? 0 : k <= 3 ? k + 1 : k == 4 ? 5 : k + 2;
            constexpr int count = (k == 0 || k == 4 || k == 8) ? 2 : 1;
            e.template sub<P, piece, first>(acc, oh, ol);
            if constexpr (count == 2) e.template sub<P, piece, first + 1>(acc, oh, ol);
        }
    }
}

// the whole epilogue of a pair at once (nothing to hide it under)
template <class Epi, int P, int NS>
__device__ __forceinline__ void epi_all(Epi& e, f32x16 (&acc)[2], u32x4 (&oh)[NS], u32x4 (&ol)[NS]) {
    static_for<96>([&](auto s_tag) { epi_slot<Epi, P, decltype(s_tag)::value, 12, NS>(e, acc, oh, ol); });
}

}  // namespace h3
}  // namespace scn
