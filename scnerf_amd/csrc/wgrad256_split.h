// wgrad256_split.h -- the 256 x 256 weight-gradient GEMMs on the bf16 matrix pipe with fp32-grade products.
//
//     dW_j[n][k] = sum_p dZ_j[p][n] * X_j[p][k],      db_j[n] = sum_p dZ_j[p][n]        (as wgrad256.h)
//
// gfx950 runs v_mfma_f32_32x32x2_f32 at 1/16 of the bf16 rate (MI355X_MICROARCH.md), so an fp32 product is
// cheaper as a sum of bf16 products.  Every fp32 operand x is cut, exactly, into three bf16 numbers
//     x = h + m + l,     h = x with the low 16 bits cleared,  m = (x - h) likewise,  l = x - h - m
// (24 significand bits = 3 x 8; both subtractions are exact), and the product a * b is accumulated in fp32 as
//     ah bh + ah bm + am bh + am bm + ah bl + al bh
// Each of the six bf16 x bf16 products is exact in fp32; the three dropped ones (am bl, al bm, al bl) are below
// 2^-23 |a b|, the size of ONE rounding of the fp32 accumulator the exact-fp32 MFMA chain performs per k-step as
// well.  Six v_mfma_f32_32x32x16_bf16 (32 cycles each) cover what takes sixteen v_mfma_f32_32x32x2_f32
// (64 cycles each): 3072 instead of 8192 matrix-pipe cycles per 16 samples of a 128 x 128 wave tile.
//
// Data path.  Operands arrive tile-native fp32 (mlp_common.h: per 32-sample tile [t][q][lane][4], the 16-byte
// piece (t, q, lane = m + 32 h) = features 32 t + 8 q + 4 h .. +3 of sample m).  A workgroup (2 x 2 waves, one per
// SIMD) owns the 256 x 256 output of one job for a contiguous chunk of samples and walks it in slabs of 16
// samples.  Staging: a thread loads 16-byte pieces (8 consecutive lanes = 8 samples of one piece column = one
// 128-byte line), cuts them (and / sub / and / sub per float, one v_perm per pair and plane) and writes the three
// planes with ds_write_b64 into an LDS image [16 samples][Ah Am Al Bh Bm Bl, 256 bf16 each].  The MFMA wants, per
// lane, 8 SAMPLES of one feature: the transpose is the LDS read, ds_read_b64_tr_b16 (each 16-lane group
// fetches a [4 samples][16 features] block and hands lane c column c).
//
// LDS image, chosen so that both access patterns are conflict-free (SQ_LDS_BANK_CONFLICT: 60 % of the LDS
// cycles with a plain padded [16][1536 + 32] image):
//   row (sample) m at m * 3104 bytes, +16 bytes for m & 4: 3104 = 32 (mod 256), so the 4 rows of a transposing
//   read sit 32 bytes apart modulo 256;
//   inside a plane the 16-feature block b = f >> 4 sits at ((b >> 1) & 3) * 32 + (b >> 3) * 256 + (b & 1) * 128
//   bytes: the two halves of a 32-feature tile, read by the two 16-lane groups of a lane half, are 128 bytes
//   apart -- a 32-lane half touches 8 distinct 32-byte bank groups;
//   a staging write of 16 lanes (8 samples x 2 pieces) covers {0,32,64,96} + {0,16} + {0,8}: all 32 write banks.
//
// Schedule.  One slab = 96 MFMA slots, products in the order (Ah Bh)(Am Bh)(Al Bh)(Ah Bm)(Ah Bl)(Am Bm) so that
// Bh is free after the third and Ah after the fifth: the next slab's Bh / Ah are read under the last two
// products.  Fillers by slot g (r = g mod 12): the 56 cut steps (8 pieces x 7) of the slab TWO ahead, written
// into LDS buffer (s + 2) mod 3, on r = 0 1 3 5 6 8 10; the 48 operand reads on r = 2 3 4 7 9 11 (8 per product =
// the plane the next product needs); the reload of the piece just cut (slab s + 4, two staging register sets:
// 2 slabs = ~6000 cycles to land) on r = 8: at most 5 instructions between two MFMAs, one barrier per slab.
//
// Measured (tools/ubench/wgrad_split_lab.hip, P = 786 432, per GEMM): 0.48 ms against 0.72 ms for wgrad256.h
// (1.5x; 215 "fp32 TFLOP/s").  The kernel is POWER-bound, not issue-bound: the matrix pipe is busy 83-90 % of
// the cycles, but the shader clock drops from 2.37 GHz (fp32 MFMA) to 1.65-1.9 GHz under bf16 MFMAs at this
// density (profiles/r02c_split_lab_pmc.txt), and neither removing the LDS conflicts nor a burstier / smoother
// filler schedule moved the time; MFMAs + operand reads alone take 0.35 ms.  v_pk_add_f32 for the cut's
// subtractions: 3 % slower.  All nine products instead of six: same error, 0.64 ms.
#pragma once
#include <type_traits>

#include <scn_wave.h>

#include "wgrad256.h"

namespace scn {
namespace wg256s {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
using wg256::Args;
using wg256::Job;

constexpr int kThreads = 256;
constexpr int kKS = 16;                        // samples per slab (the K of one bf16 MFMA)
constexpr int kW = 256;
constexpr int kPlane = kW;                     // bf16 elements of one plane of one operand in an LDS row
constexpr int kRowEl = 6 * kPlane + 16;        // 1552 bf16 = 3104 bytes = 32 (mod 256)
constexpr int kRdStep = 4 * kRowEl + 8;        // rows 4 .. 7 of every 8 sit 16 bytes to the right
constexpr int kSlabEl = kKS * kRowEl + 8;
constexpr unsigned kLdsBytes = 3u * kSlabEl * 2u;          // 149 040: three slab images
static_assert((kRowEl * 2) % 256 == 32 && (kSlabEl * 2) % 16 == 0, "row stride class of the LDS image");

enum : int {
    kNoLoad = 1,          // timing experiment: no global loads / cuts / LDS commits
    kNoBarrier = 2,       // timing experiment: no barriers (results are wrong)
    kNine = 4,            // all nine products (accuracy experiment)
};

// rank of slot g among the operand-read slots (g mod 12 in {2 3 4 7 9 11}) of its 16-slot phase
constexpr int read_rank(int g) {
    int k = 0;
    for (int x = g / 16 * 16; x < g; ++x) {
        const int r = x % 12;
        k += (r == 2 || r == 3 || r == 4 || r == 7 || r == 9 || r == 11) ? 1 : 0;
    }
    return k;
}

template <int FLAGS>
__global__ __launch_bounds__(kThreads, 1) void wgrad256_split_kernel(Args a) {
    short* lds = dynamic_lds<short>();
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const int wn = wave >> 1, wk = wave & 1;
    const Job& J = a.job[blockIdx.y];
    const long p_begin = (long)blockIdx.x * a.chunk;
    const long p_end = min(a.Ppad, p_begin + a.chunk);
    const int n_slab = p_begin < p_end ? (int)((p_end - p_begin + 31) / 32) * 2 : 0;       // always even
    float* const pw_block = J.part_w + (long)blockIdx.x * kW * kW;
    float* const pb_block = J.part_b ? J.part_b + (long)blockIdx.x * kW : nullptr;

    if (n_slab == 0) {
        for (int e = tid * 4; e < kW * kW; e += kThreads * 4)
            *reinterpret_cast<f32x4*>(pw_block + e) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (pb_block && tid < kW) pb_block[tid] = 0.f;
        return;
    }

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 bsum[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

    // ---- staging geometry: piece (fg, m): 4-feature group fg = c + 8 wave + 32 j, in-slab sample m = ml + 8 mh
    const int ml = lane & 7, c = (lane >> 3) & 7;
    const int fg0 = c + 8 * wave;
    // float offset inside a 32-sample tile block of piece (fg, sample mt): ((fg >> 1) * 64 + 32 (fg & 1) + mt) * 4
    const unsigned src0 = ((fg0 >> 1) * 64 + 32 * (fg0 & 1) + ml) * 4;       // + j * 4096 + mh * 32 + half * 64
    // LDS position (bf16 elements) of feature f inside a plane: 16-feature block b = f >> 4 at
    // ((b >> 1) & 3) * 16 + (b >> 3) * 128 + (b & 1) * 64: the two halves of a 32-feature tile are 128 bytes apart
    const int dst0 = ml * kRowEl + (ml >> 2) * 8 + wave * 16 + (c >> 2) * 64 + (c & 3) * 4;   // + mh * 8 rows + j * 128 + plane/operand
    // ---- operand reads: 16-lane group gq = lane >> 4 covers features 16 (gq & 1) .. +15, k half gq >> 1
    const int ll = lane & 15, gq = lane >> 4;
    const int rd_row = 8 * (gq >> 1) + (ll >> 2);
    const int rd_col = 64 * (gq & 1) + 4 * (ll & 3);
    const int rd_a = rd_row * kRowEl + 128 * wn + rd_col;                    // + plane * 256 + tile * 16 + rd * kRdStep
    const int rd_b = rd_row * kRowEl + 3 * kPlane + 128 * wk + rd_col;

    f32x4 raw[2][8];       // [set][operand * 4 + j * 2 + mh]
    auto load_slab = [&](auto set_tag, int s, int first, int count) {
        constexpr int SET = decltype(set_tag)::value;
        if constexpr (!(FLAGS & kNoLoad)) {
            s = min(s, n_slab - 1);
            const long p0 = p_begin + (long)(s >> 1) * 32;
            const float* bA = J.A + p0 * kW + (s & 1) * 64 + src0;
            const float* bB = J.B + p0 * kW + (s & 1) * 64 + src0;
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                if (x < first || x >= first + count) continue;
                const int o = x >> 2, j = (x >> 1) & 1, mh = x & 1;
                raw[SET][x] = load_stream(reinterpret_cast<const f32x4*>((o ? bB : bA) + j * 4096 + mh * 32));
            }
        }
    };
    // Cutting one staged piece, in seven steps small enough to sit between two MFMAs (<= 5 instructions each):
    // 0 1 3 4: element 0 1 2 3 (bias add, and / sub / and / sub);  2 5: pack the pairs (3 v_perm);  6: 3 writes
    unsigned cu[4], c1[4], c2[4];
    u32x2 ph, pm, pl;
    auto cut_step = [&](auto set_tag, int buf, auto piece_tag, auto step_tag) {
        constexpr int SET = decltype(set_tag)::value, X = decltype(piece_tag)::value, STEP = decltype(step_tag)::value;
        constexpr int o = X >> 2, j = (X >> 1) & 1, mh = X & 1;
        if constexpr (!(FLAGS & kNoLoad)) {
            if constexpr (STEP == 0 || STEP == 1 || STEP == 3 || STEP == 4) {
                constexpr int e = STEP < 2 ? STEP : STEP - 1;
                const float xe = raw[SET][X][e];
                if constexpr (o == 0) bsum[j][e] = add_raw(bsum[j][e], xe);
                cu[e] = __float_as_uint(xe);
                const float d1 = xe - __uint_as_float(cu[e] & 0xffff0000u);
                c1[e] = __float_as_uint(d1);
                const float d2 = d1 - __uint_as_float(c1[e] & 0xffff0000u);
                c2[e] = __float_as_uint(d2);
            } else if constexpr (STEP == 2 || STEP == 5) {
                constexpr int w = STEP == 2 ? 0 : 1;
                ph[w] = high_halves(cu[2 * w], cu[2 * w + 1]);
                pm[w] = high_halves(c1[2 * w], c1[2 * w + 1]);
                pl[w] = high_halves(c2[2 * w], c2[2 * w + 1]);
            } else {
                short* d = lds + buf * kSlabEl + dst0 + mh * 8 * kRowEl + j * 128 + o * 3 * kPlane;
                *reinterpret_cast<u32x2*>(d) = ph;
                *reinterpret_cast<u32x2*>(d + kPlane) = pm;
                *reinterpret_cast<u32x2*>(d + 2 * kPlane) = pl;
            }
        }
    };
    auto sync = [&]() {
        if constexpr (!(FLAGS & kNoBarrier)) block_sync();
    };

    s16x8 Ah[4], Am[4], Al[4], Bh[4], Bm[4], Bl[4];
    s16x4 half_lo;
    // read k = 0 .. 7 of a plane: tile k >> 1, samples 0-3 (k even) / 4-7 (k odd) of the lane's k-group
    auto read_piece = [&](s16x8 (&dst)[4], int buf, int base, int plane, auto k_tag) {
        constexpr int K = decltype(k_tag)::value;
        const short* s = lds + buf * kSlabEl + base + plane * kPlane + (K >> 1) * 16 + (K & 1) * kRdStep;
        if constexpr ((K & 1) == 0) {
            half_lo = lds_read_tr16(s);
        } else {
            const s16x4 hi = lds_read_tr16(s);
            const s16x8 v = {half_lo[0], half_lo[1], half_lo[2], half_lo[3], hi[0], hi[1], hi[2], hi[3]};
            dst[K >> 1] = v;
        }
    };
    auto read_plane = [&](s16x8 (&dst)[4], int buf, int base, int plane) {
        read_piece(dst, buf, base, plane, std::integral_constant<int, 0>{});
        read_piece(dst, buf, base, plane, std::integral_constant<int, 1>{});
        read_piece(dst, buf, base, plane, std::integral_constant<int, 2>{});
        read_piece(dst, buf, base, plane, std::integral_constant<int, 3>{});
        read_piece(dst, buf, base, plane, std::integral_constant<int, 4>{});
        read_piece(dst, buf, base, plane, std::integral_constant<int, 5>{});
        read_piece(dst, buf, base, plane, std::integral_constant<int, 6>{});
        read_piece(dst, buf, base, plane, std::integral_constant<int, 7>{});
    };
    // sixteen MFMAs, `filler(slot)` in front of each; JOUTER: the B tile changes slowest (B was read last)
    auto product = [&](const s16x8 (&x)[4], const s16x8 (&y)[4], auto jouter_tag, auto filler) {
        constexpr bool JOUTER = decltype(jouter_tag)::value;
        auto one = [&](auto slot_tag) {
            constexpr int S = decltype(slot_tag)::value;
            constexpr int i = JOUTER ? (S & 3) : (S >> 2), j = JOUTER ? (S >> 2) : (S & 3);
            filler(slot_tag);
            sched_fence();
            acc[i][j] = mfma_32x32x16_bf16(x[i], y[j], acc[i][j]);
            sched_fence();
        };
        one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{});
        one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
        one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{});
        one(std::integral_constant<int, 6>{}); one(std::integral_constant<int, 7>{});
        one(std::integral_constant<int, 8>{}); one(std::integral_constant<int, 9>{});
        one(std::integral_constant<int, 10>{}); one(std::integral_constant<int, 11>{});
        one(std::integral_constant<int, 12>{}); one(std::integral_constant<int, 13>{});
        one(std::integral_constant<int, 14>{}); one(std::integral_constant<int, 15>{});
    };
    using IOuter = std::false_type;
    using JOuter = std::true_type;

    // One slab = 96 MFMA slots.  Fillers by slot g (r = g mod 12): the 56 cut steps of the slab TWO ahead (8 pieces
    // x 7 steps, cut into LDS buffer (s + 2) mod 3) on r = 0 1 3 5 6 8 10; the 48 operand reads on r = 2 3 4 7 9 11
    // (8 per phase = one plane); the reload of the piece just cut (slab s + 4, same register set) on r = 8.
    // At most 5 instructions per slot.  One barrier per slab, at its end: buffer (s + 2) mod 3 was last read in
    // slab s - 1, and what is written in slab s is first read in slab s + 1 (phases 4-5: next Bh, Ah).
    // KIND 0: steady; 1: nothing left to load; 2: nothing left to cut either; 3: last slab (no next planes)
    auto slab = [&](auto set_tag, auto kind_tag, int s, int cur, int nxt, int fil) {
        constexpr int KIND = decltype(kind_tag)::value;
        auto filler = [&](auto ph_tag, s16x8 (&dst)[4], int from, int base, int plane) {
            return [&, from, base, plane](auto slot_tag) {
                constexpr int G = 16 * decltype(ph_tag)::value + decltype(slot_tag)::value;
                constexpr int R = G % 12;
                constexpr int CS = R == 0 ? 0 : R == 1 ? 1 : R == 3 ? 2 : R == 5 ? 3 : R == 6 ? 4 : R == 8 ? 5 : R == 10 ? 6 : -1;
                if constexpr (KIND < 2 && CS >= 0)
                    cut_step(set_tag, fil, std::integral_constant<int, G / 12>{}, std::integral_constant<int, CS>{});
                constexpr bool READ = R == 2 || R == 3 || R == 4 || R == 7 || R == 9 || R == 11;
                if constexpr (READ && !(KIND == 3 && G >= 64)) {
                    constexpr int K = read_rank(G);
                    read_piece(dst, from, base, plane, std::integral_constant<int, K>{});
                }
                if constexpr (KIND == 0 && R == 8) load_slab(set_tag, s + 4, G / 12, 1);
            };
        };
        product(Ah, Bh, IOuter{}, filler(std::integral_constant<int, 0>{}, Am, cur, rd_a, 1));
        product(Am, Bh, IOuter{}, filler(std::integral_constant<int, 1>{}, Al, cur, rd_a, 2));
        product(Al, Bh, IOuter{}, filler(std::integral_constant<int, 2>{}, Bm, cur, rd_b, 1));
        product(Ah, Bm, JOuter{}, filler(std::integral_constant<int, 3>{}, Bl, cur, rd_b, 2));
        if constexpr (FLAGS & kNine) {
            auto none = [](auto) {};
            product(Am, Bl, IOuter{}, none); product(Al, Bm, IOuter{}, none); product(Al, Bl, IOuter{}, none);
        }
        product(Ah, Bl, JOuter{}, filler(std::integral_constant<int, 4>{}, Bh, nxt, rd_b, 0));
        product(Am, Bm, IOuter{}, filler(std::integral_constant<int, 5>{}, Ah, nxt, rd_a, 0));
        if constexpr (KIND < 3) sync();
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    auto cut_whole = [&](auto set_tag, int buf) {
        auto whole = [&](auto piece_tag) {
            cut_step(set_tag, buf, piece_tag, std::integral_constant<int, 0>{}); cut_step(set_tag, buf, piece_tag, std::integral_constant<int, 1>{});
            cut_step(set_tag, buf, piece_tag, std::integral_constant<int, 2>{}); cut_step(set_tag, buf, piece_tag, std::integral_constant<int, 3>{});
            cut_step(set_tag, buf, piece_tag, std::integral_constant<int, 4>{}); cut_step(set_tag, buf, piece_tag, std::integral_constant<int, 5>{});
            cut_step(set_tag, buf, piece_tag, std::integral_constant<int, 6>{});
        };
        whole(std::integral_constant<int, 0>{}); whole(std::integral_constant<int, 1>{});
        whole(std::integral_constant<int, 2>{}); whole(std::integral_constant<int, 3>{});
        whole(std::integral_constant<int, 4>{}); whole(std::integral_constant<int, 5>{});
        whole(std::integral_constant<int, 6>{}); whole(std::integral_constant<int, 7>{});
    };
    load_slab(S0{}, 0, 0, 8);
    load_slab(S1{}, 1, 0, 8);
    cut_whole(S0{}, 0);
    load_slab(S0{}, 2, 0, 8);
    cut_whole(S1{}, 1);
    load_slab(S1{}, 3, 0, 8);
    sync();
    read_plane(Ah, 0, rd_a, 0);
    read_plane(Bh, 0, rd_b, 0);

    int cur = 0, nxt = 1, fil = 2;
    auto rotate = [&]() { const int t = cur; cur = nxt; nxt = fil; fil = t; };
    int s = 0;
    for (; s + 4 < n_slab; s += 2) {
        slab(S0{}, K0{}, s, cur, nxt, fil); rotate();
        slab(S1{}, K0{}, s + 1, cur, nxt, fil); rotate();
    }
    if (n_slab >= 4) {
        slab(S0{}, K1{}, s, cur, nxt, fil); rotate();
        slab(S1{}, K1{}, s + 1, cur, nxt, fil); rotate();
    }
    slab(S0{}, K2{}, n_slab - 2, cur, nxt, fil); rotate();
    slab(S1{}, K3{}, n_slab - 1, cur, nxt, fil);

    // ---- partial sums: tile (i, j) element r of lane (li = lane & 31, mh2 = lane >> 5) is
    //      dW[128 wn + 32 i + (r&3) + 8 (r>>2) + 4 mh2][128 wk + 32 j + li]
    {
        const int li = lane & 31, mh2 = lane >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 128 * wn + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * mh2;
                    pw_block[n * kW + 128 * wk + 32 * j + li] = acc[i][j][r];
                }
    }
    // bias sums: the pieces the threads staged cover every (sample, 4-feature group) of dZ once; fold the 8
    // sample lanes of a group (the two in-slab halves already share a register)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        f32x4 v = bsum[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = v[e];
            x += shfl_xor(x, 1); x += shfl_xor(x, 2); x += shfl_xor(x, 4);
            v[e] = x;
        }
        if (pb_block && ml == 0) *reinterpret_cast<f32x4*>(pb_block + 4 * (fg0 + 32 * j)) = v;
    }
}

}  // namespace wg256s
}  // namespace scn
