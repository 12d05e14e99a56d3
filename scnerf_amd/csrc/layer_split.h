// layer_split.h -- one 256 -> 256 linear layer of the network over ALL samples on the bf16 matrix pipe at fp32
// accuracy:      Z[p][n] = act( sum_k W[n][k] X[p][k] + b[n] )          (tile-native X, Z of width 256)
//
// Arithmetic as wgrad256_split.h: every fp32 number is cut exactly into three bf16 numbers (h, m, l) and a
// product is the fp32 sum of six bf16 x bf16 partial products -- (Wh Xh)(Wh Xm)(Wh Xl)(Wm Xh)(Wm Xm)(Wl Xh).
// The weights are cut ONCE per optimizer step by the packing pass; only the activations are cut here.
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
// as their registers fall free) sit between the MFMAs, at most 4 per slot; one barrier per slab.
//
// Measured (tools/ubench/layer_split_lab.hip, P = 786 432): 0.56-0.58 ms per layer against 0.80 ms for a layer inside
// the fused fp32-MFMA kernel; MFMAs + weight stream alone 0.30-0.32 ms, + activation loads and cuts 0.44, + epilogue
// 0.57.  Like wgrad256_split.h the kernel is POWER-bound (shader clock ~1.7 GHz), so what counts is the number of
// instructions, not where they sit: spreading the epilogue's 1400 instructions per block over the 144 MFMA slots
// around the block seam instead of issuing them as a burst changed nothing (0.573 vs 0.558 ms); staggering the
// workgroups' block boundaries neither.  What did matter: a bias read from global memory inside the epilogue
// waits on vmcnt, which also counts the stores just issued -- one store round trip per 16-byte piece, 20 us per
// block (the table is read from LDS instead); accumulator reads hoisted out of their pieces spill, and a scratch
// reload waits on vmcnt just the same.
// Open (profiles/r02d_layer_lab_pmc.txt): the matrix pipe is busy 46-50 % of the cycles at 2.04 GHz, 36 % of the wave
// cycles sit in s_waitcnt, so this kernel -- unlike wgrad256_split -- is stall-bound, not power-bound.  Ablations
// (lab, one box): MFMAs + weight stream 0.32 ms; + activation LOADS 0.45 (the cut's VALU on top: +0.005, i.e. hidden);
// + epilogue 0.54.  Kept from the hunt: wave-uniform load bases made opaque to the compiler (uniform_global) -- with
// 64-bit per-lane pointers every load carried ~20 VALU of address arithmetic and the compiler recycled landed
// staging registers for the temporaries behind an `s_waitcnt vmcnt(0)` once per two slabs (0.575 -> 0.54 ms); the
// weight loads issued BEFORE the slab's activation loads and waited for three iterations later (vmcnt counts in
// order: a weight wait covers every older activation load; neutral by itself).  Tried, no effect: two independent
// dependency chains per cut step; non-temporal activation loads; buffer-descriptor loads (they bloat the code to
// 11 k lines through loop unswitching and slow the weight stream).  What the activation loads cost is not yet
// explained (they are issued two slabs = 4 us ahead and the waits are counted, vmcnt(19)); with them the L1 reports
// pending-miss stalls 68 % of the cycles (TCP_PENDING_STALL, 6 % without), but halving the L1 requests -- four waves
// side by side along the samples, each owning all 256 features of 64 samples, so that no two waves load the same
// activations -- moved the cost into the epilogue (32 dependent pieces per tile) and the total not at all (0.547 ms).
// Per CU and slab 56 KB go through the L1's 64 B/clk address unit (896 of ~4000 cycles), four lock-stepped waves at
// a time; giving every wave load slots of its own (uniform `if (wave == k)` around each load) made it WORSE
// (0.77 ms): behind every such branch the compiler's wait-count pass falls back to `s_waitcnt vmcnt(0)` (156 of them).
// Two rows of the epilogue interleaved with prefetched bias pieces: no change (the epilogue's cost is its 64 stores
// per wave, 0.056 ms, and 0.035 ms of arithmetic).  Run on half of the CUs each workgroup is 25 % faster (MFMAs
// + weight stream alone: 21 %): the full chip is also clock-limited (2.04 GHz sustained).  Without any barrier in the
// slab loop (racy, timing only): 0.551 vs 0.555 ms -- neither the barrier nor the lock-step of the waves costs.
#pragma once
#include <type_traits>

#include <scn_wave.h>

namespace scn {
namespace lsp {

constexpr int kThreads = 256;
constexpr int kSlabs = 16;                          // K = 256 in slabs of 16
constexpr int kSlabUnits = 3 * 8 * 64;              // 16-byte fragments of one slab image
constexpr int kSlabShorts = kSlabUnits * 8;
constexpr unsigned kLdsBytes = 2u * kSlabUnits * 16u + 1024u;  // two slab images + the bias table

struct Args {
    const float* X;        // tile-native, width 256: K slabs 0 .. 15
    const float* X2;       // row-major [Ppad][x2_ld] (the skip layer's encoded points): K slabs 16 .. n_k - 1, or nullptr
    int x2_ld;
    int n_k;               // K slabs per block: 16, or 16 + x2 width / 16 (even)
    const short* W;        // [n_k][3][8][64][8] bf16 bits (pack_planes)
    const float* bias;     // lane-vector table of 8 tiles: entry ((4 t + q) * 2 + h) * 4 + j
    float* Z;              // tile-native, width 256
    unsigned* mask;        // ReLU bits [wave tile][64 lanes][4 words] or nullptr
    long Ppad;             // samples covered (multiple of 128)
    int relu;
    // mode 1 (data gradients): Z = select(mask_in, W X + bias[n] * vec[p]) -- the ReLU gate of the layer below and
    // the rank-1 density-head term (bias = alpha_linear's weights, vec = d sigma); no bias add, no bits written
    int mode;
    const unsigned* mask_in;   // [wave tile][64 lanes][4 words]
    const float* vec;          // per-sample scalar, element p * vec_stride (p < n_vec), or nullptr
    int vec_stride;
    long n_vec;
};

enum : int {
    kNoEpilogue = 1,      // timing experiment: only the last block is stored
    kNoCut = 2,           // timing experiment: no loads / cuts of X (planes hold garbage)
    kNoCutMath = 8,       // timing experiment: X is loaded but not cut (the raw words serve as planes)
    kNoBarrier = 64,      // timing experiment: no workgroup barriers in the slab loop (racy: results are wrong)
    kPlainStore = 4,      // experiment: default-policy stores instead of non-temporal ones (no difference)
    kNoZStore = 16,       // experiment: the epilogue computes but stores only the mask words
};

template <int N> using I = std::integral_constant<int, N>;

template <int FLAGS>
__global__ __launch_bounds__(kThreads, 1) void layer_split_kernel(Args a) {
    short* lds = dynamic_lds<short>();
    float* const lds_bias = reinterpret_cast<float*>(lds + 2 * kSlabShorts);
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const int wn = wave >> 1, wp = wave & 1;
    const int m = lane & 31, g = lane >> 5;
    const long n_tiles = a.Ppad / 32;
    const long n_blocks = (n_tiles + 7) / 8;
    if ((long)blockIdx.x >= n_blocks) return;
    const int my_blocks = (int)((n_blocks - blockIdx.x + gridDim.x - 1) / gridDim.x);
    const int n_k = a.n_k;

    f32x16 acc[4][4];       // written by MFMAs only: a block's first product starts from the constant 0

    // ---- X: lane (m, g) of sample tile j, slab s: pieces ((2 s + g) * 64 + m + 32 h') * 4 floats, h' = 0, 1
    // Every load is `global_load v, v_off, s[base]`: a wave-uniform base (uniform_global: opaque to the compiler) + a
    // 32-bit per-lane byte offset.  With 64-bit per-lane pointers each load cost ~20 VALU of address arithmetic and the
    // compiler recycled landed staging registers for the temporaries behind an `s_waitcnt vmcnt(0)`.
    const int wp_u = uniform(wp);
    const unsigned xoff_a = (unsigned)((g * 64 + m) * 16);                                   // main operand, piece h' = 0
    const unsigned xoff_b = (unsigned)((m * a.x2_ld + 8 * g) * 4);                           // second operand
    auto tile_of = [&](int b, int j) {             // sample tile j of this workgroup's b-th block, clamped into the tensor
        const long blk = blockIdx.x + (long)(b < my_blocks ? b : my_blocks - 1) * gridDim.x;
        const long t = blk * 8 + wp_u * 4 + j;
        return t < n_tiles ? t : n_tiles - 1;
    };
    f32x4 raw[2][4][2];                             // [set = slab parity][sample tile][h']
    // (b, s): block and K slab; s >= 16: the row-major second operand, k = 16 (s - 16) + 8 g + 4 h' .. +3
    auto load_x = [&](auto set_tag, int b, int s, auto j_tag) {
        constexpr int SET = decltype(set_tag)::value, j = decltype(j_tag)::value;
        if constexpr (!(FLAGS & kNoCut)) {
            const long t = tile_of(b, j);
            // (selects, not a branch: control flow here would cut the MFMA stream into scheduling regions)
            const bool main_part = s < 16;
            const float* p1 = a.X + t * 8192 + s * 512;
            const float* p2 = a.X2 + t * 32 * a.x2_ld + (s - 16) * 16;
            const global_bytes base = uniform_global(main_part ? p1 : p2);
            const unsigned off0 = main_part ? xoff_a : xoff_b;
            const unsigned off1 = off0 + (main_part ? 512u : 16u);
            raw[SET][j][0] = load_f32x4(base, off0);      // (non-temporal loads: no difference; the two waves that
            raw[SET][j][1] = load_f32x4(base, off1);      // share a sample tile meet in L1 either way)
        }
    };
    // planes of the fragment being used / being cut: [ping-pong][plane]
    s16x8 xp[2][3];
    unsigned cu[8], c1[8], c2[8];
    // step 0..7: element e (and / sub / and / sub); 8, 9, 10: pack plane h, m, l (4 v_perm each)
    auto cut_step = [&](auto set_tag, auto j_tag, auto dst_tag, auto step_tag) {
        constexpr int SET = decltype(set_tag)::value, j = decltype(j_tag)::value, D = decltype(dst_tag)::value,
                      STEP = decltype(step_tag)::value;
        if constexpr (FLAGS & kNoCutMath) {
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

    // ---- W: slab image copy (6 x 16 bytes per thread) and fragment reads
    f32x4 wst[6];
    const unsigned woff = (unsigned)tid * 16u;
    auto load_w = [&](int s, auto x_tag) {
        constexpr int x = decltype(x_tag)::value;
        wst[x] = load_f32x4(uniform_global(a.W + (long)s * kSlabShorts) + x * (kThreads * 16), woff);
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
        one(I<0>{}); one(I<1>{}); one(I<2>{}); one(I<3>{}); one(I<4>{}); one(I<5>{}); one(I<6>{}); one(I<7>{});
        one(I<8>{}); one(I<9>{}); one(I<10>{}); one(I<11>{}); one(I<12>{}); one(I<13>{}); one(I<14>{}); one(I<15>{});
        one(I<16>{}); one(I<17>{}); one(I<18>{}); one(I<19>{}); one(I<20>{}); one(I<21>{}); one(I<22>{}); one(I<23>{});
    };

    // slab u out of LDS buffer BUF (= u & 1 = raw set): per sample tile j
    //   slots 0-10   cut the next fragment ((u, j + 1), or (u + 1, 0) out of the other raw set)
    //   slot 11 (j = 0: slot 23)   reload raw fragment j for the slab two ahead (load_x: both pieces)
    //   j = 0: slots 13-16 read Wl(u);   slots 17-22 load W(u + 2) into the staging registers
    //   j = 2: barrier at the end (every wave is done with W(u): its buffer is free; W(u + 1), written during slab
    //          u - 1, is visible)
    //   j = 3: slots 0-5 write W(u + 2) into buffer BUF;   slots 12-15 read Wh(u + 1);   slots 20-23 read Wm(u + 1)
    // vmcnt counts loads IN ORDER: waiting for a weight piece also waits for every activation load issued before it.
    // The weight loads therefore go out BEFORE slab u's activation loads and are waited for three iterations later,
    // when the youngest activation load ahead of them is a whole slab old (with the weight loads in iterations 2-3
    // and their writes right after, every slab waited twice for activation loads two iterations old).
    auto slab = [&](auto buf_tag, auto first_tag, int b, int s) {
        constexpr int BUF = decltype(buf_tag)::value;
        using Set = I<BUF>;
        using Other = I<BUF ^ 1>;
        const bool wraps = s + 2 >= n_k;               // two slabs ahead: the next block's slab 0 / 1
        const int b2 = wraps ? b + 1 : b, s2 = wraps ? s + 2 - n_k : s + 2;
        auto fill = [&](auto j_tag) {
            return [&](auto slot_tag) {
                constexpr int j = decltype(j_tag)::value, S = decltype(slot_tag)::value;
                if constexpr (j == 3 && S <= 5) write_w(BUF, I<S>{});
                if constexpr (S <= 10) {
                    if constexpr (j < 3) cut_step(Set{}, I<j + 1>{}, I<(j + 1) & 1>{}, I<S>{});
                    else cut_step(Other{}, I<0>{}, I<0>{}, I<S>{});
                }
                if constexpr (S == (j == 0 ? 23 : 11)) load_x(Set{}, b2, s2, I<j>{});
                if constexpr (j == 0) {
                    if constexpr (S >= 13 && S <= 16) read_w(BUF, I<2>{}, I<S - 13>{});
                    if constexpr (S >= 17 && S <= 22) load_w(s2, I<S - 17>{});
                } else if constexpr (j == 3) {
                    if constexpr (S >= 12 && S <= 15) read_w(BUF ^ 1, I<0>{}, I<S - 12>{});
                    if constexpr (S >= 20) read_w(BUF ^ 1, I<1>{}, I<S - 20>{});
                }
            };
        };
        iteration(I<0>{}, I<0>{}, first_tag, fill(I<0>{}));
        iteration(I<1>{}, I<1>{}, first_tag, fill(I<1>{}));
        iteration(I<2>{}, I<0>{}, first_tag, fill(I<2>{}));
        sync();
        iteration(I<3>{}, I<1>{}, first_tag, fill(I<3>{}));
    };

    // finished block: bias, activation, ReLU bits, tile-native stores
    auto epilogue = [&](int b) {
        {
            const long blk = blockIdx.x + (long)b * gridDim.x;
            const float lo = a.relu ? 0.f : -__builtin_huge_valf();
            const long tile0 = blk * 8 + uniform(wp) * 4;              // wave-uniform: the range checks are scalar branches
            auto tile = [&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
                const long t = tile0 + j;
                if (t >= n_tiles) return;
                float* zt = a.Z + t * 8192 + wn * 4096 + lane * 4;      // + (4 i + q) * 256
                unsigned bits[2] = {0u, 0u};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned hb = 0u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // (the bias piece is re-read for every tile: kept in registers across the four sample tiles
                        // it costs 64 VGPRs the slab loop does not have.  From LDS, not from memory: a global load
                        // here waits on vmcnt, which counts the stores of the previous piece too -- one store round
                        // trip per piece, 20 us per block.  The fence keeps the accumulator reads piece by piece:
                        // hoisted, they spill, and a scratch reload waits on vmcnt just the same.)
                        sched_fence();
                        const f32x4 b = *reinterpret_cast<const f32x4*>(lds_bias + ((4 * (4 * wn + i) + q) * 2 + g) * 4);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = max_raw(add_raw(acc[i][j][4 * q + e], b[e]), lo);
                            hb = shift_in_positive(hb, v[e]);
                        }
                        f32x4* dst = reinterpret_cast<f32x4*>(zt + (4 * i + q) * 256);
                        if constexpr (FLAGS & kNoZStore) { (void)dst; }
                        else if constexpr (FLAGS & kPlainStore) *dst = v;
                        else __builtin_nontemporal_store(v, dst);
                    }
                    bits[i >> 1] = (bits[i >> 1] << 16) | hb;
                }
                if (a.mask) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<u32x2*>(a.mask + t * 256 + lane * 4 + 2 * wn) = u32x2{bits[0], bits[1]};
                }
            };
            // data-gradient form: gate by the ReLU bits of the layer below (element 16 t + r of a lane: word t >> 1,
            // bit 31 - (16 (t & 1) + r)), after adding the density head's rank-1 term
            auto tile_bwd = [&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
                const long t = tile0 + j;
                if (t >= n_tiles) return;
                float* zt = a.Z + t * 8192 + wn * 4096 + lane * 4;
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 gate = *reinterpret_cast<const u32x2*>(a.mask_in + t * 256 + lane * 4 + 2 * wn);
                const long p = t * 32 + m;
                const float vs = (a.vec && p < a.n_vec) ? a.vec[p * a.vec_stride] : 0.f;
                // RANK1: the density head's term (feature_linear^T only); the other seven layers skip the FMA and
                // the LDS read of the table
                auto tile_rows = [&](auto rank1_tag) {
                    constexpr bool RANK1 = decltype(rank1_tag)::value;
                    auto row = [&](auto i_tag) {
                        constexpr int i = decltype(i_tag)::value;
                        auto piece = [&](auto q_tag) {
                            constexpr int q = decltype(q_tag)::value;
                            sched_fence();
                            f32x4 b = {0.f, 0.f, 0.f, 0.f};
                            if constexpr (RANK1) b = *reinterpret_cast<const f32x4*>(lds_bias + ((4 * (4 * wn + i) + q) * 2 + g) * 4);
                            constexpr int bit0 = 31 - (16 * (i & 1) + 4 * q);
                            const unsigned word = gate[i >> 1];
                            auto val = [&](int e) { return RANK1 ? fmaf(b[e], vs, acc[i][j][4 * q + e]) : acc[i][j][4 * q + e]; };
                            f32x4 v;
                            v[0] = keep_if_bit<bit0 - 0>(val(0), word);
                            v[1] = keep_if_bit<bit0 - 1>(val(1), word);
                            v[2] = keep_if_bit<bit0 - 2>(val(2), word);
                            v[3] = keep_if_bit<bit0 - 3>(val(3), word);
                            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(zt + (4 * i + q) * 256));
                        };
                        piece(I<0>{}); piece(I<1>{}); piece(I<2>{}); piece(I<3>{});
                    };
                    row(I<0>{}); row(I<1>{}); row(I<2>{}); row(I<3>{});
                };
                if (a.vec) tile_rows(std::true_type{}); else tile_rows(std::false_type{});
            };
            if (a.mode == 1) {
                tile_bwd(I<0>{}); tile_bwd(I<1>{}); tile_bwd(I<2>{}); tile_bwd(I<3>{});
                return;
            }
            tile(I<0>{}); tile(I<1>{}); tile(I<2>{}); tile(I<3>{});
        }
    };

    lds_bias[tid] = a.bias[tid];
    // ---- prologue: W slabs 0 and 1 -> buffers 0 and 1; X raw of slabs 0 and 1; first fragment cut
    load_w(0, I<0>{}); load_w(0, I<1>{}); load_w(0, I<2>{}); load_w(0, I<3>{}); load_w(0, I<4>{}); load_w(0, I<5>{});
    load_x(I<0>{}, 0, 0, I<0>{}); load_x(I<0>{}, 0, 0, I<1>{}); load_x(I<0>{}, 0, 0, I<2>{}); load_x(I<0>{}, 0, 0, I<3>{});
    load_x(I<1>{}, 0, 1, I<0>{}); load_x(I<1>{}, 0, 1, I<1>{}); load_x(I<1>{}, 0, 1, I<2>{}); load_x(I<1>{}, 0, 1, I<3>{});
    write_w(0, I<0>{}); write_w(0, I<1>{}); write_w(0, I<2>{}); write_w(0, I<3>{}); write_w(0, I<4>{}); write_w(0, I<5>{});
    load_w(1, I<0>{}); load_w(1, I<1>{}); load_w(1, I<2>{}); load_w(1, I<3>{}); load_w(1, I<4>{}); load_w(1, I<5>{});
    write_w(1, I<0>{}); write_w(1, I<1>{}); write_w(1, I<2>{}); write_w(1, I<3>{}); write_w(1, I<4>{}); write_w(1, I<5>{});
    sync();
    read_w(0, I<0>{}, I<0>{}); read_w(0, I<0>{}, I<1>{}); read_w(0, I<0>{}, I<2>{}); read_w(0, I<0>{}, I<3>{});
    read_w(0, I<1>{}, I<0>{}); read_w(0, I<1>{}, I<1>{}); read_w(0, I<1>{}, I<2>{}); read_w(0, I<1>{}, I<3>{});
    {
        auto whole = [&](auto s) { cut_step(I<0>{}, I<0>{}, I<0>{}, s); };
        whole(I<0>{}); whole(I<1>{}); whole(I<2>{}); whole(I<3>{}); whole(I<4>{}); whole(I<5>{}); whole(I<6>{});
        whole(I<7>{}); whole(I<8>{}); whole(I<9>{}); whole(I<10>{});
    }

    for (int b = 0; b < my_blocks; ++b) {
        slab(I<0>{}, std::true_type{}, b, 0);
        slab(I<1>{}, std::false_type{}, b, 1);
        for (int s = 2; s < n_k; s += 2) {
            slab(I<0>{}, std::false_type{}, b, s);
            slab(I<1>{}, std::false_type{}, b, s + 1);
        }
        if constexpr (!(FLAGS & kNoEpilogue)) epilogue(b);
    }
    if constexpr (FLAGS & kNoEpilogue) epilogue(my_blocks - 1);       // (keeps the MFMAs alive)
}

}  // namespace lsp
}  // namespace scn
