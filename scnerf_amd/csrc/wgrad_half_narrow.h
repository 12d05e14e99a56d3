// wgrad_half_narrow.h -- the narrower weight-gradient GEMMs of a network pass on THREE fp16 products per product.
//
//     dW[n][k] = sum_p dZ[p][n] * X[p][k]      n < WA, k < WB,      db[n] = sum_p dZ[p][n]
//
//   WA x WB        A (dZ)                    B (X)                          reference layer (run_nerf_helpers.py:92-128)
//   256 x  64      trunk dZ, tile-native     encoded points, row-major      pts_linears[0], skip part of pts_linears[5]
//   256 x 128      trunk dZ, tile-native     encoded 4-D points, row-major  the same layers of NeRF++'s background net
//   128 x 256      views-layer dZ, tile-nat. feature, tile-native           views_linears[0] (feature part)
//
// wgrad_tiles.h runs these shapes on the fp32 MFMA, where they are bound by that pipe (97-110 TFLOP/s: 2.1 ms of a
// 13.6 ms step).  Here the arithmetic, the LDS image and the transposing reads are wgrad256_half.h's (operands scaled
// by one power of two per workgroup, cut into two fp16 numbers, products (Ah Bh) + (Al Bh) + (Ah Bl), fp32
// accumulate); with a quarter / half of that kernel's MFMAs per slab the launch is no longer bound by the matrix
// pipe but by reading its operands, so the slab loop is left to the compiler's scheduler (no slot tables): per slab
// of 16 samples a thread issues the LDS reads of its planes of the current slab, cuts its 3 .. 6 staged pieces of the
// NEXT slab into LDS, refills those staging registers (four sets: the loads run four slabs ahead of the cut) and issues
// 12 / 24 MFMAs, product-major; three LDS slab images, one barrier per slab.  Measured at P = 786 432: 256 x 64 0.20 ms
// (5.0 TB/s of operand bytes; 0.265 ms on the fp32 MFMA), 128 x (256 + 32) 0.29 ms (0.47 + 0.19 ms in two launches).
//
// Scales.  A workgroup owns a contiguous range of samples; its scale per operand comes from the chunk maxima of the
// resident kernels (mlp_fwd_h3_kernel.h, mlp_bwd_h3_kernel.h: [row][coarse chunk]) -- the largest over the coarse
// chunks its range touches, optionally through an affine bound (feature = W h7 + b is bounded by the feature layer's
// largest row 1-norm x max |h7| + its largest |bias|; the scale table of the packing pass has both).  A maximum that is
// too LARGE only costs low-order bits of the small values (wgrad256_half.h).
//
// A row-major B operand may hold anything in rows >= P: those rows are staged as zeros (as wgrad_tiles.h).
//
// Second X operand (WB2 = 32; the views layer: X = [feature | encoded direction]).  The 128 x 32 product against the
// encoded direction shares dZ with the 128 x 256 one, so it rides along instead of reading dZ again: its 32 columns
// (row-major [P][32]) take the unused half of the LDS row's A region (WA = 128 fills tiles 0 .. 3; tile 4 holds them),
// waves 0 and 1 stage them, and the waves with wk = 0 multiply them with their A tiles into TA more accumulators.
#pragma once
#include <type_traits>

#include <scn_wave.h>

#include "wgrad256_half.h"

namespace scn {
namespace wgnh {

using wg256h::kKS;
using wg256h::kPlane;
using wg256h::kRdStep;
using wg256h::kRowEl;
using wg256h::kSlabEl;
using wg256h::kThreads;
using wg256h::u32x2;
// Two slab images: slab s + 1 is cut into the image slab s - 1 was read from, which every wave has left behind the
// barrier that ended iteration s - 1 (wgrad256_half.h keeps three because its cut runs two slabs ahead).  66 KB instead of
// 99: with at most 256 registers per wave TWO workgroups fit a CU, and one's dependent chain of a slab -- LDS reads, cut,
// MFMAs, barrier -- runs under the other's.
constexpr int kImages = 2;
constexpr unsigned kLdsBytes = (unsigned)kImages * kSlabEl * 2u;
constexpr int kSets = 4;                       // staging register sets = slabs the loads run ahead

struct Bound {
    const float* amax;     // [n_coarse] chunk maxima of the operand (or of what bounds it)
    const float* mul;      // nullptr, or a device float: bound = amax * mul + add
    const float* add;
};

struct Args {
    const float* A;        // dZ, tile-native, width WA
    const float* B;        // X: tile-native width WB, or row-major [P][WB]
    float* part_w;         // [G][WA][WB]
    float* part_b;         // [G][WA] or nullptr
    long P;                // valid samples
    long Ppad;             // samples the tile-native sections cover (multiple of 128)
    long chunk;            // samples per workgroup (multiple of 32)
    Bound a, b;
    int n_coarse;          // chunks of the maxima
    long coarse_chunk;     // samples per chunk of the maxima
    // second X operand (WB2 > 0): row-major [P][WB2], partials [G][WA][WB2]
    const float* B2;
    float* part_w2;
    Bound b2;
    // a second GEMM of the same shape on the same X (blockIdx.y == 1; grid (G, 2)): layer 0 and the skip columns of layer 5
    // both multiply the encoded point -- launched together their workgroups pair up on a CU and read X once from HBM
    const float* A_y1;
    float* part_w_y1;
    float* part_b_y1;
    Bound a_y1;
};

// where the 4-feature group fg of an operand sits in its 256-position region of an LDS row (wgrad256_half.h's image)
__device__ __forceinline__ int region_pos(int fg) {
    return 128 * (fg >> 5) + 16 * ((fg >> 3) & 3) + 64 * ((fg & 7) >> 2) + 4 * (fg & 3);
}

template <int WA, int WB, bool B_ROWMAJOR, int WB2 = 0>
__global__ __launch_bounds__(kThreads, (WA == 256 && WB == 64 && WB2 == 0) ? 2 : 1) void wgrad_half_narrow_kernel(Args a) {
    if constexpr (!(WA == 256 && WB == 64 && WB2 == 0)) claim_whole_register_file();     // the one-wave-per-SIMD shapes admit no guest (scn_wave.h)
    if (blockIdx.y == 1) { a.A = a.A_y1; a.part_w = a.part_w_y1; a.part_b = a.part_b_y1; a.a = a.a_y1; }
    constexpr int TA = WA / 64, TB = WB / 64;             // accumulator tiles per wave
    constexpr int PA = WA / 64, PB = WB / 64;             // 16-byte pieces per thread and slab
    constexpr int P2 = WB2 ? 1 : 0;                       // the second X operand's piece (waves 0 and 1 only)
    static_assert((TA == 2 || TA == 4) && (TB == 1 || TB == 2 || TB == 4), "wave tile shapes");
    static_assert(WB2 == 0 || (WB2 == 32 && WA == 128), "the second X operand sits in the unused half of the A region");
    short* lds = dynamic_lds<short>();
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const int wn = wave >> 1, wk = wave & 1;
    const long p_begin = (long)blockIdx.x * a.chunk;
    const long p_end = min(a.Ppad, p_begin + a.chunk);
    const int n_slab = p_begin < p_end ? (int)((p_end - p_begin + 31) / 32) * 2 : 0;       // always even
    float* const pw_block = a.part_w + (long)blockIdx.x * WA * WB;
    float* const pb_block = a.part_b ? a.part_b + (long)blockIdx.x * WA : nullptr;

    float* const pw2_block = WB2 ? a.part_w2 + (long)blockIdx.x * WA * (WB2 ? WB2 : 1) : nullptr;

    if (n_slab == 0) {
        for (int e = tid * 4; e < WA * WB; e += kThreads * 4)
            *reinterpret_cast<f32x4*>(pw_block + e) = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (WB2 > 0)
            for (int e = tid * 4; e < WA * WB2; e += kThreads * 4)
                *reinterpret_cast<f32x4*>(pw2_block + e) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (pb_block && tid < WA) pb_block[tid] = 0.f;
        return;
    }
    // ---- scales: the largest maximum over the coarse chunks [p_begin, p_end) touches
    auto bound_of = [&](const Bound& b) {
        const int c0 = (int)(p_begin / a.coarse_chunk);
        const int c1 = min(a.n_coarse - 1, (int)((p_end - 1) / a.coarse_chunk));
        float v = 0.f;
        for (int c = min(c0, a.n_coarse - 1); c <= c1; ++c) v = fmaxf(v, b.amax[c]);
        if (b.mul) v = __builtin_fmaf(v, *b.mul, *b.add);
        return v;
    };
    const float sa = wg256h::scale_for(bound_of(a.a));
    const float sb = wg256h::scale_for(bound_of(a.b));
    float sb2 = 1.f;
    if constexpr (WB2 > 0) sb2 = wg256h::scale_for(bound_of(a.b2));

    f32x16 acc[TA][TB];
    f32x16 acc2[WB2 ? TA : 1];
    if constexpr (WB2 > 0) {
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 bsum[PA / 2];
#pragma unroll
    for (int q = 0; q < PA / 2; ++q) bsum[q] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging geometry.  Piece q of an operand with NP pieces per thread: 4-feature group fg, in-slab sample m
    //   NP 4: fg = c + 8 wave + 32 (q >> 1), m = ml + 8 (q & 1)        (wgrad256_half.h)
    //   NP 2: fg = c + 8 wave,               m = ml + 8 q
    //   NP 1: fg = c + 8 (wave & 1),         m = ml + 8 (wave >> 1)
    const int ml = lane & 7, c = (lane >> 3) & 7;
    auto piece_fg = [&](auto np_tag, int q) {
        constexpr int NP = decltype(np_tag)::value;
        return NP == 4 ? c + 8 * wave + 32 * (q >> 1) : (NP == 2 ? c + 8 * wave : c + 8 * (wave & 1));
    };
    auto piece_m = [&](auto np_tag, int q) {
        constexpr int NP = decltype(np_tag)::value;
        return NP == 4 ? ml + 8 * (q & 1) : (NP == 2 ? ml + 8 * q : ml + 8 * (wave >> 1));
    };
    using NPA = std::integral_constant<int, PA>;
    using NPB = std::integral_constant<int, PB>;
    const int ll = lane & 15, gq = lane >> 4;
    const int rd_row = 8 * (gq >> 1) + (ll >> 2);
    const int rd_col = 64 * (gq & 1) + 4 * (ll & 3);
    // wave (wn, i) owns A tile ta = wn TA + i at region position 128 (ta >> 2) + 16 (ta & 3); B likewise behind 2 planes
    auto tile_pos = [](int t) { return 128 * (t >> 2) + 16 * (t & 3); };
    const int rd_a = rd_row * kRowEl + rd_col;
    const int rd_b = rd_row * kRowEl + 2 * kPlane + rd_col;

    f32x4 raw[kSets][PA + PB + P2];
    auto load_slab = [&](auto set_tag, int s) {
        constexpr int SET = decltype(set_tag)::value;
        s = min(s, n_slab - 1);
        const long p0 = p_begin + (long)(s >> 1) * 32;
        const int m0 = 16 * (s & 1);
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int fg = piece_fg(NPA{}, q), m = piece_m(NPA{}, q) + m0;
            raw[SET][q] = load_stream(reinterpret_cast<const f32x4*>(a.A + p0 * WA + ((fg >> 1) * 64 + 32 * (fg & 1) + m) * 4));
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int fg = piece_fg(NPB{}, q), m = piece_m(NPB{}, q) + m0;
            if constexpr (B_ROWMAJOR) {
                const long p = p0 + m;
                raw[SET][PA + q] = p < a.P ? load_stream(reinterpret_cast<const f32x4*>(a.B + p * WB + 4 * fg)) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                raw[SET][PA + q] = load_stream(reinterpret_cast<const f32x4*>(a.B + p0 * WB + ((fg >> 1) * 64 + 32 * (fg & 1) + m) * 4));
            }
        }
        if constexpr (WB2 > 0) {
            // waves 0, 1: 4-column group c of in-slab sample ml + 8 wave
            if (wave < 2) {
                const long p = p0 + m0 + ml + 8 * wave;
                raw[SET][PA + PB] = p < a.P ? load_stream(reinterpret_cast<const f32x4*>(a.B2 + p * WB2 + 4 * c)) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto cut_piece = [&](const f32x4& x4, float s, short* d) {
        u32x2 ph, pl;
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
            ph[w2] = pack_f16_scaled(x4[2 * w2], x4[2 * w2 + 1], s);
            pl[w2] = pack_f16(residual_f16<0>(x4[2 * w2], s, ph[w2]), residual_f16<1>(x4[2 * w2 + 1], s, ph[w2]));
        }
        *reinterpret_cast<u32x2*>(d) = ph;
        *reinterpret_cast<u32x2*>(d + kPlane) = pl;
    };
    auto cut_slab = [&](auto set_tag, int buf) {
        constexpr int SET = decltype(set_tag)::value;
        short* const img = lds + buf * kSlabEl;
#pragma unroll
        for (int x = 0; x < PA + PB; ++x) {
            const bool is_b = x >= PA;
            const int q = is_b ? x - PA : x;
            const int fg = is_b ? piece_fg(NPB{}, q) : piece_fg(NPA{}, q);
            const int m = is_b ? piece_m(NPB{}, q) : piece_m(NPA{}, q);
            const float s = is_b ? sb : sa;
            const f32x4 x4 = raw[SET][x];
            if (!is_b) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bsum[PA == 4 ? (q >> 1) : 0][e] = add_raw(bsum[PA == 4 ? (q >> 1) : 0][e], x4[e]);
            }
            cut_piece(x4, s, img + m * kRowEl + ((m & 7) >> 2) * 8 + region_pos(fg) + (is_b ? 2 * kPlane : 0));
        }
        if constexpr (WB2 > 0) {
            if (wave < 2) {
                const int m = ml + 8 * wave;
                cut_piece(raw[SET][PA + PB], sb2, img + m * kRowEl + ((m & 7) >> 2) * 8 + region_pos(32 + c));
            }
        }
    };
    // one plane of one 32-feature tile -> the MFMA operand (8 samples of the lane's k-group)
    auto read_tile = [&](int buf, int base, int plane, int tile) {
        const short* s = lds + buf * kSlabEl + base + plane * kPlane + tile_pos(tile);
        const s16x4 lo = lds_read_tr16(s);
        const s16x4 hi = lds_read_tr16(s + kRdStep);
        return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    // One slab.  Source order = issue order: the LDS reads of the current slab first, then the cut of the next slab (its
    // ~100 VALU instructions cover the reads' latency; its LDS writes queue behind the reads), then the MFMAs, whose
    // operands have arrived by then.
    auto step_body = [&](auto next_set_tag, int s, int cur, int nxt) {
        s16x8 Ah[TA], Al[TA], Bh[TB], Bl[TB], Ch, Cl;
#pragma unroll
        for (int j = 0; j < TB; ++j) { Bh[j] = read_tile(cur, rd_b, 0, wk * TB + j); Bl[j] = read_tile(cur, rd_b, 1, wk * TB + j); }
#pragma unroll
        for (int i = 0; i < TA; ++i) { Ah[i] = read_tile(cur, rd_a, 0, wn * TA + i); Al[i] = read_tile(cur, rd_a, 1, wn * TA + i); }
        if constexpr (WB2 > 0) {
            if (wk == 0) { Ch = read_tile(cur, rd_a, 0, 4); Cl = read_tile(cur, rd_a, 1, 4); }
        }
        // the NEXT slab: registers -> image `nxt` (last read in iteration s - 2: every wave has passed the barrier that
        // ended it), then its set is refilled five slabs ahead
        if (s + 1 < n_slab) {
            cut_slab(next_set_tag, nxt);
            load_slab(next_set_tag, s + 5);
        }
        // product-major: the three products of one accumulator tile are TA TB MFMAs apart (back to back they would wait
        // out each other's latency: 64 cycles per dependent 32-cycle MFMA)
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) acc[i][j] = mfma_32x32x16_f16(Ah[i], Bh[j], acc[i][j]);
        if constexpr (WB2 > 0) {
            if (wk == 0) {
#pragma unroll
                for (int i = 0; i < TA; ++i) acc2[i] = mfma_32x32x16_f16(Ah[i], Ch, acc2[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) acc[i][j] = mfma_32x32x16_f16(Al[i], Bh[j], acc[i][j]);
        if constexpr (WB2 > 0) {
            if (wk == 0) {
#pragma unroll
                for (int i = 0; i < TA; ++i) acc2[i] = mfma_32x32x16_f16(Al[i], Ch, acc2[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) acc[i][j] = mfma_32x32x16_f16(Ah[i], Bl[j], acc[i][j]);
        if constexpr (WB2 > 0) {
            if (wk == 0) {
#pragma unroll
                for (int i = 0; i < TA; ++i) acc2[i] = mfma_32x32x16_f16(Ah[i], Cl, acc2[i]);
            }
        }
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    // slab k waits in set k mod 4; slab s is cut into image s mod kImages during iteration s - 1 and read during iteration s
    load_slab(S0{}, 0);
    load_slab(S1{}, 1);
    load_slab(S2{}, 2);
    load_slab(S3{}, 3);
    cut_slab(S0{}, 0);
    load_slab(S0{}, 4);
    block_sync();
    int cur = 0, nxt = 1;
    auto step = [&](auto next_set_tag, int s) {
        step_body(next_set_tag, s, cur, nxt);
        block_sync();
        cur = nxt;
        nxt = nxt == kImages - 1 ? 0 : nxt + 1;
    };
    for (int s = 0; s < n_slab; s += 4) {
        step(S1{}, s);
        if (s + 1 < n_slab) step(S2{}, s + 1);
        if (s + 2 < n_slab) step(S3{}, s + 2);
        if (s + 3 < n_slab) step(S0{}, s + 3);
    }

    // ---- partial sums, un-scaled: tile (i, j) element r of lane (li, mh2) is dW[32 (wn TA + i) + (r&3) + 8 (r>>2) + 4 mh2][32 (wk TB + j) + li]
    {
        const float una = __uint_as_float(0x7f000000u - __float_as_uint(sa)), unb = __uint_as_float(0x7f000000u - __float_as_uint(sb));
        const int li = lane & 31, mh2 = lane >> 5;
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 32 * (wn * TA + i) + (r & 3) + 8 * (r >> 2) + 4 * mh2;
                    pw_block[n * WB + 32 * (wk * TB + j) + li] = (acc[i][j][r] * una) * unb;
                }
        if constexpr (WB2 > 0) {
            if (wk == 0) {
                const float unb2 = __uint_as_float(0x7f000000u - __float_as_uint(sb2));
#pragma unroll
                for (int i = 0; i < TA; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = 32 * (wn * TA + i) + (r & 3) + 8 * (r >> 2) + 4 * mh2;
                        pw2_block[n * WB2 + li] = (acc2[i][r] * una) * unb2;
                    }
            }
        }
    }
    // bias sums: a thread's pieces of one feature group over its 2 in-slab samples and all slabs; 8 sample lanes to fold
#pragma unroll
    for (int jq = 0; jq < PA / 2; ++jq) {
        f32x4 v = bsum[jq];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = v[e];
            x += shfl_xor(x, 1); x += shfl_xor(x, 2); x += shfl_xor(x, 4);
            v[e] = x;
        }
        if (pb_block && ml == 0) *reinterpret_cast<f32x4*>(pb_block + 4 * (c + 8 * wave + 32 * jq)) = v;
    }
}

}  // namespace wgnh
}  // namespace scn
