// mlp_h3p.h -- LAB (tools/ubench/residency_lab.hip; never part of the product build): the resident arithmetic of mlp_h3.h in its PAIRED residency: 16 samples per wave on
// v_mfma_f32_16x16x32_f16, eight waves per workgroup = TWO waves per SIMD at <= 256 registers.
//
// Why.  mlp_h3.h's wave owns 32 samples x 256 features: two operand buffers of 128 registers each, ~490 of the 512 a lone
// wave per SIMD has.  One in-order wave cannot issue to the matrix pipe while one of its memory instructions sits in the
// CU's vector-memory path (~60-100 cycles each: the weight stream's loads, their LDS writes, the activation stores), so
// the matrix pipe's 1.70 ms and the memory path's ~0.9 ms per launch ADD (profiles/r03_ablation_h3.txt,
// r05_ablation_h3.txt).  Half the samples per wave halve every per-wave array: two operand buffers 2 x 64 registers,
// accumulator pairs 2 x 8, and a wave fits 256 -- the SIMD then holds a second instruction stream whose MFMAs issue while
// the first one's memory instruction waits.  The workgroup still takes 128 samples through the weight stream (8 waves x
// 16), so the L2 -> LDS bytes per sample are unchanged.
//
// The price is LDS -> register fragment traffic: a 16 x 16 x 32 MFMA does half the MACs of a 32 x 32 x 16 one per 1 KB A
// fragment, so a layer reads 2 MB of fragments per CU where mlp_h3.h reads 1 MB: 170 B per clock at the matrix pipe's
// rate.  ds_read_b128 moves 256 B per clock per CU on gfx950 (MI355X_MICROARCH.md, LDS table; round 4's arithmetic took
// 128 and called this residency LDS-bound) -- 0.67 of the LDS array, plus the stream's writes.  Measured:
// profiles/r05_lab_residency.txt (tools/ubench/residency_lab.hip runs a chain of eight 256 -> 256 layers in both
// residencies from the product's building blocks and these).
//
// Shape.  Wave = 16 samples x all features.  The product is computed transposed, D[feature][sample]: lane (m = l & 15,
// g = l >> 4) owns feature 16 T + 4 g + j of sample m in register j of output tile T (16 features).  The MFMA contracts
// 32 k per instruction, 8 per lane group: the two accumulators (2 s, 2 s + 1) of a lane ARE its 8 elements of K slab s of
// the next layer -- element e of lane group g is feature 32 s + 16 (e >> 2) + 4 g + (e & 3) -- once the weights are packed
// in that order (tools/residency_lab.py: fragment_index).  Output tiles are produced pair by pair as in mlp_h3.h: a pair = one K slab of the
// next layer, its epilogue (two pieces of four registers) dealt out under the next pair's 48 MFMA slots.
//
// Weight stream: mlp_h3.h's format -- 16-byte A fragments in consumption order, a unit = [Wh T0][Wh T1][Wl T0][Wl T1] of
// one K slab = six MFMAs, 32 KB chunks of 8 units through three LDS buffers, one barrier per chunk -- with 512 threads
// moving a chunk in four pieces each instead of eight.
//
// Activations and gradients are stored in mlp_common.h's TILE-NATIVE layout unchanged (per 32-sample tile a block
// [t][q][lane32][4]): the two waves of a 32-sample tile each write 4 runs of 256 B per store instruction instead of one run
// of 1 KB, so the weight-gradient GEMMs and the other residency's kernels read what they always read.
#pragma once
#include "mlp_h3.h"

namespace scn {

#ifndef SCNERF_SIMT_EMU_BUILD
// v_mfma_f32_16x16x32_f16: D[16x16] += A[16x32] * B[32x16], 16 cycles/SIMD (half the MACs of the 32x32x16 form per 1 KB
// operand).  Lane l supplies A[i = l&15][k = 8 (l>>4) .. +7] and B[k = 8 (l>>4) .. +7][j = l&15]; it receives column
// j = l&15 of D, rows 4 (l>>4) + r for r = 0..3.
__device__ __forceinline__ f32x4 mfma_16x16x32_f16(s16x8 a, s16x8 b, f32x4 c) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
#else
// v_mfma_f32_16x16x32_f16: lane l gives A[l&15][8 (l>>4) .. +7], B[8 (l>>4) .. +7][l&15]; receives column l&15, rows 4 (l>>4) + r
inline f32x4 mfma_16x16x32_f16(s16x8 a, s16x8 b, f32x4 c) {
    uint64_t mine[4];
    std::memcpy(&mine[0], &a, 16);
    std::memcpy(&mine[2], &b, 16);
    uint64_t all[4][64];
    for (int w = 0; w < 4; ++w) {
        const uint64_t* x = simt::wave_exchange(mine[w]);
        std::memcpy(all[w], x, sizeof(all[w]));
    }
    auto elem = [&](int word0, int lane, int k) {
        const uint64_t u = all[word0 + (k >> 2)][lane];
        return f16_value((unsigned)(uint16_t)(u >> (16 * (k & 3))));
    };
    const int l = simt::cur_lane();
    const int j = l & 15, g = l >> 4;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        double sum = 0.0;
        for (int k = 0; k < 32; ++k) sum += (double)elem(0, i + 16 * (k >> 3), k & 7) * (double)elem(2, j + 16 * (k >> 3), k & 7);
        c[r] = (float)((double)c[r] + sum);
    }
    return c;
}
#endif

namespace h3p {

using namespace scn::mlp;
using scn::h3::I;
using scn::h3::NoFill;
using scn::h3::as_frag;
using scn::h3::cut8;
using scn::h3::inv_pow2;
using scn::h3::kChunkBytes;
using scn::h3::kStreamLds;
using scn::h3::scale_for;
using scn::h3::static_for;
using scn::h3::u32x4;

constexpr int kThreadsP = 512;             // 8 waves, two per SIMD
constexpr int kSamplesPerWaveP = 16;
constexpr int kPieceBytes = kThreadsP * 16;        // what one instruction of the workgroup moves: 8 KB
constexpr int kPieces = kChunkBytes / kPieceBytes; // 4 per chunk and thread

// ---- weight stream ------------------------------------------------------------------------------------------------
struct Stream {
    global_bytes g;        // the chunk to fetch next (two ahead of the one in use)
    unsigned cur;          // LDS byte offset of the buffer in use: 0, 32 K, 64 K
    f32x4 stage[kPieces];
    __device__ __forceinline__ unsigned next() const { return cur == 2u * kChunkBytes ? 0u : cur + kChunkBytes; }
};

// first chunk -> buffer 0, second chunk -> staging registers; the caller synchronises before the first fragment read
__device__ __forceinline__ void stream_prime(Stream& ws, const void* stream, char* lds, unsigned tid16) {
    ws.g = uniform_global(stream);
#pragma unroll
    for (int i = 0; i < kPieces; ++i) ws.stage[i] = load_f32x4(ws.g + i * kPieceBytes, tid16);
#pragma unroll
    for (int i = 0; i < kPieces; ++i) *reinterpret_cast<f32x4*>(lds + i * kPieceBytes + tid16) = ws.stage[i];
    ws.g = uniform_global(ws.g + kChunkBytes);
#pragma unroll
    for (int i = 0; i < kPieces; ++i) ws.stage[i] = load_f32x4(ws.g + i * kPieceBytes, tid16);
    ws.g = uniform_global(ws.g + kChunkBytes);
    ws.cur = 0u;
    if constexpr (lab::kNoStream) {        // (timing experiments: the stream is off, every buffer holds real fragments)
#pragma unroll
        for (int b = 1; b < 3; ++b)
#pragma unroll
            for (int i = 0; i < kPieces; ++i) *reinterpret_cast<f32x4*>(lds + b * kChunkBytes + i * kPieceBytes + tid16) = ws.stage[i];
    }
}

// slot KAPPA (0 .. 47) of a chunk: piece i of the NEXT chunk goes from its staging register to LDS in slot 6 i and the
// register is refilled in slot 6 i + 1 with piece i of the chunk after that (a load is waited for 47 slots after its
// issue); the barrier at slot 30: behind it the next chunk is complete (its first fragments are read in slots 42 .. 45)
// and every wave is done with the chunk before this one (whose buffer the writes of the NEXT chunk's slots 0 .. 18 reuse).
template <int KAPPA>
__device__ __forceinline__ void stream_slot(Stream& ws, char* lds, unsigned tid16) {
    if constexpr (!lab::kNoStream && KAPPA < 24 && KAPPA % 6 == 0)
        *reinterpret_cast<f32x4*>(lds + ws.next() + (KAPPA / 6) * kPieceBytes + tid16) = ws.stage[KAPPA / 6];
    if constexpr (!lab::kNoBarrier && KAPPA == 30) block_sync();
    constexpr int LOAD_PIECE = lab::kNoStream ? -1 : ((KAPPA < 24 && KAPPA % 6 == 1) ? KAPPA / 6 : -1);
    if constexpr (LOAD_PIECE >= 0)
        ws.stage[LOAD_PIECE] = load_f32x4(uniform_global(ws.g + LOAD_PIECE * kPieceBytes), pinned_here(tid16));
    if constexpr (KAPPA == 47) {
        ws.cur = ws.next();
        ws.g = uniform_global(ws.g + kChunkBytes);
    }
}

struct Wave {
    char* lds;
    unsigned tid16, lane16;
    Stream ws;
    s16x8 ring[2][4];      // A fragments of the unit in use / of the next unit
};

__device__ __forceinline__ void ring_prime(Wave& w) {
#pragma unroll
    for (int j = 0; j < 4; ++j) w.ring[0][j] = *reinterpret_cast<const s16x8*>(w.lds + j * 1024 + w.lane16);
}

// One UNIT of the stream: four fragments, six MFMAs (U = the unit's index in the whole stream: place in the chunk U % 8,
// ring half U & 1).
//   PAIR:   one K slab of an output-tile pair -- fragments [Wh0 Wh1 Wl0 Wl1]:
//           (Wh0 Xh)(Wh1 Xh)(Wh0 Xl)(Wh1 Xl)(Wl0 Xh)(Wl1 Xh), tile 0 into acc[0], tile 1 into acc[1]
//   !PAIR:  two K slabs A, B of ONE output tile -- fragments [WhA WlA WhB WlB]; slab A into acc[0], slab B into acc[1]
// `fill(I<j>)` is issued in front of MFMA j; slots 0 .. 3 also fetch the NEXT unit's fragments.
template <int U, bool PAIR, bool FIRST, class Fill>
__device__ __forceinline__ void unit(Wave& w, u32x4 xhA, u32x4 xlA, u32x4 xhB, u32x4 xlB, f32x4 (&acc)[2], Fill&& fill) {
    constexpr int PH = U & 7, RP = U & 1;
    static_for<6>([&](auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        stream_slot<PH * 6 + j>(w.ws, w.lds, w.tid16);
        fill(j_tag);
        if constexpr (j < 4) {
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
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            acc[x] = mfma_16x16x32_f16(w.ring[RP][fr], as_frag(b), zero);
        } else {
            acc[x] = mfma_16x16x32_f16(w.ring[RP][fr], as_frag(b), acc[x]);
        }
        sched_fence();
    });
}

// All K slabs of one output-tile pair: NK units starting at stream unit U0.  operand(I<s>, &xh, &xl) names the planes of
// K slab s; fill(I<sigma>) is the filler of slot sigma = 6 s + j.
template <int U0, int NK, class Operand, class Fill>
__device__ __forceinline__ void tile_pair(Wave& w, f32x4 (&acc)[2], Operand&& operand, Fill&& fill) {
    static_for<NK>([&](auto s_tag) {
        constexpr int s = decltype(s_tag)::value;
        u32x4 xh, xl;
        operand(s_tag, xh, xl);
        unit<U0 + s, true, s == 0>(w, xh, xl, xh, xl, acc, [&](auto j_tag) { fill(I<6 * s + decltype(j_tag)::value>{}); });
    });
}

// All K slabs (2 NU of them) of a single output tile; the result is acc[0] + acc[1].
template <int U0, int NU, class Operand, class Fill>
__device__ __forceinline__ void tile_single(Wave& w, f32x4 (&acc)[2], Operand&& operand, Fill&& fill) {
    static_for<NU>([&](auto u_tag) {
        constexpr int u = decltype(u_tag)::value;
        u32x4 xhA, xlA, xhB, xlB;
        operand(I<2 * u>{}, xhA, xlA);
        operand(I<2 * u + 1>{}, xhB, xlB);
        unit<U0 + u, false, u == 0>(w, xhA, xlA, xhB, xlB, acc, [&](auto j_tag) { fill(I<6 * u + decltype(j_tag)::value>{}); });
    });
}

// ---- epilogue slices ------------------------------------------------------------------------------------------------
// An epilogue works on one output-tile pair (tiles 2 P, 2 P + 1 = K slab P of the next layer) in two PIECES -- tile x: its
// four accumulator registers = one 16-byte run of the tile-native section = half of the slab's lane operand -- and a piece
// in twelve SUB-steps of at most four VALU instructions.  epi_slot<.., SIGMA, SPP> maps filler slot SIGMA of the covering
// tile pair to sub-steps, SPP slots per piece (24: the 48 slots of an 8-slab pair; 18: the pair that must be done before
// the next layer's K loop reaches slab 7 at slot 42).
constexpr int kSubSteps = 12;
template <class Epi, int P, int SIGMA, int SPP, int NS>
__device__ __forceinline__ void epi_slot(Epi& e, f32x4 (&acc)[2], u32x4 (&oh)[NS], u32x4 (&ol)[NS]) {
    if constexpr (SIGMA < 2 * SPP) {
        constexpr int piece = SIGMA / SPP, k = SIGMA % SPP;
        constexpr int first = kSubSteps * k / SPP, last = kSubSteps * (k + 1) / SPP;
        static_for<last - first>([&](auto d_tag) { e.template sub<P, piece, first + decltype(d_tag)::value>(acc, oh, ol); });
    }
}

// the whole epilogue of a pair at once (nothing to hide it under)
template <class Epi, int P, int NS>
__device__ __forceinline__ void epi_all(Epi& e, f32x4 (&acc)[2], u32x4 (&oh)[NS], u32x4 (&ol)[NS]) {
    static_for<2 * kSubSteps>([&](auto s_tag) { epi_slot<Epi, P, decltype(s_tag)::value, kSubSteps, NS>(e, acc, oh, ol); });
}

// Where a lane's 16-byte piece of output tile T goes inside the 32-sample tile's block of a tile-native section
// (mlp_common.h: [t][q][lane32 = (m32, h)][4] holds feature 32 t + 8 q + 4 h + j of sample m32): feature 16 T + 4 g + j is
// t = T >> 1, q = 2 (T & 1) + (g >> 1), h = g & 1; the wave's samples are m32 = 16 (wave & 1) + m.
__device__ __forceinline__ unsigned tile_native_lane_offset(int wave, int lane) {
    const int m = lane & 15, g = lane >> 4;
    return (unsigned)((g >> 1) * 1024 + (32 * (g & 1) + 16 * (wave & 1) + m) * 16);
}
__host__ __device__ constexpr int tile_native_piece_offset(int T) { return 2 * T * 1024; }

}  // namespace h3p
}  // namespace scn
