// wgrad256_half.h -- the 256 x 256 weight-gradient GEMMs on THREE fp16 products per product.
//
//     dW_j[n][k] = sum_p dZ_j[p][n] * X_j[p][k],      db_j[n] = sum_p dZ_j[p][n]        (as wgrad256.h)
//
// wgrad256_split.h cuts every operand exactly into three bf16 numbers and spends six matrix-pipe products per fp32
// product; with those the launch is bound by the matrix pipe at the clock the power limit allows (79 % busy at
// 1.6 GHz, 3.4 ms for the eight GEMMs of the fine pass).  Here an operand is scaled by a power of two and cut into TWO
// fp16 numbers, x S = h + l (|l| <= 2^-11 |h|), and a product is (Ah Bh) + (Ah Bl) + (Al Bh) on
// v_mfma_f32_32x32x16_f16 -- the dropped (Al Bl) is 2^-22 of the product, what the six-product scheme drops too: half
// the matrix-pipe work, after which the launch is bound by reading its operands (12.9 GB) once.
//
// Scales.  The contraction runs over SAMPLES, so a scale must be common to all samples a workgroup sums over: one
// power of two per (job, operand, workgroup chunk), S = 2^k with max|x| S < 2^13 over the chunk.  The maxima come from
// the kernels that produced the operands: the resident forward / data-gradient kernels (mlp_fwd_h3.hip,
// mlp_bwd_h3.hip) know every sample's maximum when a layer ends (they scale by it themselves) and leave the largest of
// a wave's 32 samples in Args::amax with one atomic max per wave and layer -- [operand][job][chunk] floats, zeroed
// before the pass.  A value far below its chunk's maximum is carried with an ABSOLUTE error of 2^-38 of that maximum
// (fp16 subnormals of the low plane) instead of a relative 2^-22: its product with the other operand is smaller than
// the chunk's largest term by the same factor, so the sum stays at fp32 grade (checked against fp64 with per-sample
// magnitudes spread over 2^40: tests).  The partial sums of a chunk leave the kernel un-scaled, fp32, as before.
//
// Data path as wgrad256_split.h (tile-native fp32 pieces -> staging registers -> cut -> LDS image [16 samples][Ah Al
// Bh Bl] -> transposing reads ds_read_b64_tr_b16), one slab of 16 samples = 48 MFMA slots:
//   cut of the slab two ahead: 8 pieces x 6 steps (bias sums | h of pair 0 | l of pair 0 | h of pair 1 | l of pair 1 |
//   two LDS writes + the reload for four slabs ahead) = one step per slot;
//   operand reads: products in the order (Ah Bh)(Al Bh)(Ah Bl); the A planes of the NEXT slab go into the other of two
//   register sets at any time, Bl during the first two products, the next Bh under the third.
//
// Tried and reverted: alpha_linear's vector-matrix product (sum_p d sigma[p] X[p][k], X = the activation section
// feature_linear's GEMM stages anyway) riding on the X pieces' idle cut step -- four fma per piece and two 4-byte loads
// per slab.  The launch went from 2.03 to 2.34 ms (3.35 ms with 64-bit per-lane addresses for those loads), more than
// the 0.12 ms the separate HBM-rate pass costs: in a kernel scheduled slot by slot every extra load re-times the
// vmcnt waits of the operand stream.
#pragma once
#include <type_traits>
#include <utility>

#include <scn_wave.h>

#include "wgrad256.h"

namespace scn {
namespace wg256h {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
using wg256::Job;

constexpr int kThreads = 256;
constexpr int kKS = 16;                        // samples per slab (the K of one fp16 MFMA)
constexpr int kW = 256;
constexpr int kPlane = kW;                     // fp16 elements of one plane of one operand in an LDS row
constexpr int kRowEl = 4 * kPlane + 16;        // 1040 fp16 = 2080 bytes = 32 (mod 256): wgrad256_split.h's bank pattern
constexpr int kRdStep = 4 * kRowEl + 8;        // rows 4 .. 7 of every 8 sit 16 bytes to the right
constexpr int kSlabEl = kKS * kRowEl + 8;
constexpr unsigned kLdsBytes = 3u * kSlabEl * 2u;          // 99 888: three slab images
static_assert((kRowEl * 2) % 256 == 32 && (kSlabEl * 2) % 16 == 0, "row stride class of the LDS image");

struct Args {
    Job job[wg256::kMaxJobs];
    int n_jobs;
    long Ppad;             // samples the tile-native sections cover (multiple of 128)
    long chunk;            // samples per workgroup (multiple of 32)
    const float* amax_a;   // [n_jobs][gridDim.x] largest |dZ| of the chunk
    const float* amax_b;   // [n_jobs][gridDim.x] largest |X| of the chunk
};

// 2^k with bound 2^k < 2^13 (as mlp_h3.h's scale_for)
__device__ __forceinline__ float scale_for(float bound) {
    const unsigned e = (__float_as_uint(bound) >> 23) & 0xffu;
    return __uint_as_float((266u - (e < 13u ? 13u : e)) << 23);
}

// ---- which slot of a slab (0 .. 47) carries which operand reads -----------------------------------------------------
// read k of a slab (0 .. 31): 0-7 Bl of this slab (needed at slot 32), 8-15 Ah and 16-23 Al of the NEXT slab (other
// register set), 24-31 Bh of the next slab (Bh is free from slot 32 on).  Slots with g % 6 in {1, 3, 5} carry one read;
// the first cut step of an X piece (pieces 4 .. 7: no bias sum there) carries two.
constexpr int reads_in_slot(int g) {
    const int r = g % 6;
    if (r == 1 || r == 3 || r == 5) return 1;
    return (r == 0 && g >= 24) ? 2 : 0;
}
constexpr int reads_before(int g) {
    int k = 0;
    for (int x = 0; x < g; ++x) k += reads_in_slot(x);
    return k;
}
static_assert(reads_before(48) == 32 && reads_before(32) == 20, "read slots of a slab");
// the reads of slots < 32 are, in order, Bl (8), next Ah (8), next Al (4); those of slots >= 32: next Bh (8) first -- it
// is needed at the next slab's first slot -- then the rest of next Al (4)
constexpr int read_item(int k) {      // -> plane code * 8 + piece:  0 Bl, 1 next Ah, 2 next Al, 3 next Bh
    if (k < 8) return 0 * 8 + k;
    if (k < 16) return 1 * 8 + (k - 8);
    if (k < 20) return 2 * 8 + (k - 16);
    if (k < 28) return 3 * 8 + (k - 20);
    return 2 * 8 + 4 + (k - 28);
}

template <class F, int... Gs>
__device__ __forceinline__ void for_each_slot(F&& f, std::integer_sequence<int, Gs...>) {
    (f(std::integral_constant<int, Gs>{}), ...);
}

template <int FLAGS>
__global__ __launch_bounds__(kThreads, 1) void wgrad256_half_kernel(Args a) {
    claim_whole_register_file();               // 444 registers left room for a 64-register guest wave (scn_wave.h)
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
    const float sa = scale_for(a.amax_a[blockIdx.y * gridDim.x + blockIdx.x]);
    const float sb = scale_for(a.amax_b[blockIdx.y * gridDim.x + blockIdx.x]);

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 bsum[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

    // ---- staging geometry (wgrad256_split.h): piece (fg, m): 4-feature group fg = c + 8 wave + 32 j, in-slab sample
    // m = ml + 8 mh
    const int ml = lane & 7, c = (lane >> 3) & 7;
    const int fg0 = c + 8 * wave;
    const unsigned src0 = ((fg0 >> 1) * 64 + 32 * (fg0 & 1) + ml) * 4;       // + j * 4096 + mh * 32 + half * 64
    const int dst0 = ml * kRowEl + (ml >> 2) * 8 + wave * 16 + (c >> 2) * 64 + (c & 3) * 4;   // + mh * 8 rows + j * 128 + plane/operand
    const int ll = lane & 15, gq = lane >> 4;
    const int rd_row = 8 * (gq >> 1) + (ll >> 2);
    const int rd_col = 64 * (gq & 1) + 4 * (ll & 3);
    const int rd_a = rd_row * kRowEl + 128 * wn + rd_col;                    // + plane * 256 + tile * 16 + rd * kRdStep
    const int rd_b = rd_row * kRowEl + 2 * kPlane + 128 * wk + rd_col;

    f32x4 raw[2][8];       // [set][operand * 4 + j * 2 + mh]
    auto load_slab = [&](auto set_tag, int s, int first, int count) {
        constexpr int SET = decltype(set_tag)::value;
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
    };
    // Cutting one staged piece in six steps of at most four instructions:
    // 0: bias sums (dZ pieces);  1 / 3: the h plane of elements (0, 1) / (2, 3);  2 / 4: their l plane;  5: two LDS writes
    u32x2 ph, pl;
    auto cut_step = [&](auto set_tag, int buf, auto piece_tag, auto step_tag) {
        constexpr int SET = decltype(set_tag)::value, X = decltype(piece_tag)::value, STEP = decltype(step_tag)::value;
        constexpr int o = X >> 2, j = (X >> 1) & 1, mh = X & 1;
        const float s = o ? sb : sa;
        const f32x4& x4 = raw[SET][X];
        if constexpr (STEP == 0) {
            if constexpr (o == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bsum[j][e] = add_raw(bsum[j][e], x4[e]);
            }
        } else if constexpr (STEP == 1 || STEP == 3) {
            constexpr int w = STEP == 1 ? 0 : 1;
            ph[w] = pack_f16_scaled(x4[2 * w], x4[2 * w + 1], s);
        } else if constexpr (STEP == 2 || STEP == 4) {
            constexpr int w = STEP == 2 ? 0 : 1;
            pl[w] = pack_f16(residual_f16<0>(x4[2 * w], s, ph[w]), residual_f16<1>(x4[2 * w + 1], s, ph[w]));
        } else {
            short* d = lds + buf * kSlabEl + dst0 + mh * 8 * kRowEl + j * 128 + o * 2 * kPlane;
            *reinterpret_cast<u32x2*>(d) = ph;
            *reinterpret_cast<u32x2*>(d + kPlane) = pl;
        }
    };
    auto sync = [&]() { block_sync(); };

    s16x8 Ah[2][4], Al[2][4], Bh[4], Bl[4];
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

    // One slab = 48 MFMA slots = three products of sixteen.  PAR = parity of the slab = the register set its A planes
    // sit in.  KIND 0: steady; 1: nothing left to load; 2: nothing left to cut either; 3: last slab (no next planes).
    // One barrier per slab, at its end: buffer (s + 2) mod 3 was last read in slab s - 1, and what is cut during slab s
    // is first read during slab s + 1 (the planes of slab s + 2).
    auto slab = [&](auto par_tag, auto kind_tag, int s, int cur, int nxt, int fil) {
        constexpr int PAR = decltype(par_tag)::value, KIND = decltype(kind_tag)::value;
        using Set = std::integral_constant<int, PAR>;
        auto slot = [&](auto g_tag) {
            constexpr int G = decltype(g_tag)::value;
            // ---- fillers ----
            if constexpr (KIND < 2) cut_step(Set{}, fil, std::integral_constant<int, G / 6>{}, std::integral_constant<int, G % 6>{});
            if constexpr (KIND == 0 && G % 6 == 5) load_slab(Set{}, s + 4, G / 6, 1);
            constexpr int NR = reads_in_slot(G), K0 = reads_before(G);
            auto one_read = [&](auto k_tag) {
                constexpr int IT = read_item(decltype(k_tag)::value), PL = IT >> 3, PC = IT & 7;
                if constexpr (PL == 0) read_piece(Bl, cur, rd_b, 1, std::integral_constant<int, PC>{});
                else if constexpr (KIND == 3) { /* no next slab */ }
                else if constexpr (PL == 1) read_piece(Ah[PAR ^ 1], nxt, rd_a, 0, std::integral_constant<int, PC>{});
                else if constexpr (PL == 2) read_piece(Al[PAR ^ 1], nxt, rd_a, 1, std::integral_constant<int, PC>{});
                else read_piece(Bh, nxt, rd_b, 0, std::integral_constant<int, PC>{});
            };
            if constexpr (NR >= 1) one_read(std::integral_constant<int, K0>{});
            if constexpr (NR >= 2) one_read(std::integral_constant<int, K0 + 1>{});
            sched_fence();
            // ---- the MFMA: products (Ah Bh) i-outer, (Al Bh) i-outer, (Ah Bl) j-outer (Bl was read last) ----
            constexpr int PH = G / 16, S = G % 16;
            constexpr int i = PH == 2 ? (S & 3) : (S >> 2), j = PH == 2 ? (S >> 2) : (S & 3);
            if constexpr (PH == 0) acc[i][j] = mfma_32x32x16_f16(Ah[PAR][i], Bh[j], acc[i][j]);
            else if constexpr (PH == 1) acc[i][j] = mfma_32x32x16_f16(Al[PAR][i], Bh[j], acc[i][j]);
            else acc[i][j] = mfma_32x32x16_f16(Ah[PAR][i], Bl[j], acc[i][j]);
            sched_fence();
        };
        for_each_slot(slot, std::make_integer_sequence<int, 48>{});
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
    read_plane(Ah[0], 0, rd_a, 0);
    read_plane(Al[0], 0, rd_a, 1);
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

    // ---- partial sums, un-scaled: tile (i, j) element r of lane (li = lane & 31, mh2 = lane >> 5) is
    //      dW[128 wn + 32 i + (r&3) + 8 (r>>2) + 4 mh2][128 wk + 32 j + li]
    {
        // (two factors: the product of the inverse scales may itself underflow)
        const float una = __uint_as_float(0x7f000000u - __float_as_uint(sa)), unb = __uint_as_float(0x7f000000u - __float_as_uint(sb));
        const int li = lane & 31, mh2 = lane >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 128 * wn + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * mh2;
                    pw_block[n * kW + 128 * wk + 32 * j + li] = (acc[i][j][r] * una) * unb;
                }
    }
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

}  // namespace wg256h
}  // namespace scn
