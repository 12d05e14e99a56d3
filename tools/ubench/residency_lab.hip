// residency_lab.hip -- LAB (never part of the product build): a chain of eight 256 -> 256 ReLU layers on three fp16
// products per product with the activations register-resident, weight stream, epilogue (bias, ReLU, mask bits, maximum,
// cut) and training stores included, in the candidate residencies of DESIGN.md section 8, built from the product's own
// building blocks:
//   kind 0  "h3"   today's: 32 samples per wave on v_mfma_f32_32x32x16_f16, 4 waves per workgroup, one per SIMD (mlp_h3.h)
//   kind 1  "h3p"  16 samples per wave on v_mfma_f32_16x16x32_f16, 8 waves per workgroup, two per SIMD (mlp_h3p.h)
// (+ 2 for the inference instantiation: no stores, no mask words.)  The third candidate -- activations' cut planes in LDS --
// is residency_lds_lab.hip.
// Built as a shared library by tools/ubench/build_residency_lab.sh, driven by tools/residency_lab.py (numpy + ctypes: packs
// the weights, checks the last layer's output against fp64, times, and runs under rocprofv3 --pmc); the same source is
// compiled for the CPU SIMT interpreter to check the packing and the schedule's data flow without a GPU.
#include <hip/hip_runtime.h>

#include <scn_lab.h>
#include <scn_wave.h>

#include "mlp_fwd_h3_kernel.h"
#include "mlp_h3p.h"

namespace lab_chain {

using namespace scn;
using namespace scn::mlp;
using scn::h3::I;
using scn::h3::u32x4;

constexpr int kLayers = 8;
constexpr int kSections = kLayers + 1;          // section 8: the input's tiles that arrive through the pending epilogue

__host__ __device__ inline float lab_input(unsigned p, unsigned f) {
    unsigned x = p * 1103515245u + f * 12345u + p * f * 7u + 0x9e3779b9u;
    x ^= x >> 15;
    x *= 0x2c1b3c6du;
    x ^= x >> 12;
    return (float)(x >> 8) * (1.f / 16777216.f);
}

// ---- kind 1: the paired residency ----------------------------------------------------------------------------------------
template <bool ON> struct MaskBits { unsigned bits, words[2]; };
template <> struct MaskBits<false> {};

template <bool TRAIN>
struct ReluEpiP : MaskBits<TRAIN> {
    float os, s_next, am;
    const float* bias;         // LDS table of the layer, natural order, + 4 g
    global_bytes_rw save;      // the 32-sample tile's block of the layer's section (TRAIN)
    unsigned lane_store;
    f32x4 bq;
    float v[4];
    unsigned hp;

    template <int P, int PIECE, int SUB, int NS>
    __device__ __forceinline__ void sub(f32x4 (&acc)[2], u32x4 (&oh)[NS], u32x4 (&ol)[NS]) {
        constexpr int x = PIECE, T = 2 * P + x, c0 = 2 * x;
        static_assert(P < NS, "operand buffer too small for this pair");
        if constexpr (lab::kNoEpilogue) {
            if constexpr (SUB == 1) {
                oh[P][c0] = __float_as_uint(acc[x][0]) & 0x3bff3bffu; oh[P][c0 + 1] = __float_as_uint(acc[x][1]) & 0x3bff3bffu;
                ol[P][c0] = __float_as_uint(acc[x][2]) & 0x3bff3bffu; ol[P][c0 + 1] = __float_as_uint(acc[x][3]) & 0x3bff3bffu;
            }
        } else if constexpr (SUB == 0) {
            bq = *reinterpret_cast<const f32x4*>(bias + 16 * T);
            if constexpr (TRAIN && PIECE == 0 && (P & 3) == 0) this->bits = 0u;
        } else if constexpr (SUB <= 4) {
            constexpr int e = SUB - 1;
            v[e] = relu_raw(__builtin_fmaf(acc[x][e], os, bq[e]));
            if constexpr (TRAIN) this->bits = shift_in_positive(this->bits, v[e]);
        } else if constexpr (SUB == 5) {
            am = max3(am, v[0], v[1]);
            am = max3(am, v[2], v[3]);
        } else if constexpr (SUB == 6) {
            hp = pack_f16_scaled(v[0], v[1], s_next);
        } else if constexpr (SUB == 7) {
            oh[P][c0] = hp;
            ol[P][c0] = pack_f16(residual_f16<0>(v[0], s_next, hp), residual_f16<1>(v[1], s_next, hp));
        } else if constexpr (SUB == 8) {
            hp = pack_f16_scaled(v[2], v[3], s_next);
        } else if constexpr (SUB == 9) {
            oh[P][c0 + 1] = hp;
            ol[P][c0 + 1] = pack_f16(residual_f16<0>(v[2], s_next, hp), residual_f16<1>(v[3], s_next, hp));
        } else if constexpr (SUB == 10) {
            if constexpr (TRAIN && !lab::kNoStore)
                store_stream_at(uniform_global_rw(save + h3p::tile_native_piece_offset(T)), pinned_here(lane_store),
                                f32x4{v[0], v[1], v[2], v[3]});
        } else {
            if constexpr (TRAIN && PIECE == 1 && (P & 3) == 3) this->words[P >> 2] = this->bits;
        }
    }
};

constexpr unsigned kLdsP = h3::kStreamLds + kSections * 256 * 4;

template <bool TRAIN>
__global__ __launch_bounds__(h3p::kThreadsP, 2) void chain16_kernel(const short* __restrict__ wstream, const float* __restrict__ bias,
                                                                    const float* __restrict__ sc, float* __restrict__ save_arg,
                                                                    unsigned* __restrict__ out, long P) {
    using namespace scn::h3p;
    float* const save = TRAIN ? save_arg : nullptr;
    const long Ppad = padded_samples(P);
    Wave w;
    w.lds = dynamic_lds<char>();
    float* const tables = reinterpret_cast<float*>(w.lds + kStreamLds);
    for (int i = threadIdx.x; i < kSections * 256; i += kThreadsP) tables[i] = i < kLayers * 256 ? bias[i] : 0.f;
    const int lane = lane_id(), m = lane & 15, g = lane >> 4;
    const int wv = uniform(wave_id());
    const long tile32 = (long)blockIdx.x * 4 + (wv >> 1);
    const unsigned p = (unsigned)(tile32 * 32 + (wv & 1) * 16 + m);
    w.tid16 = threadIdx.x * 16u;
    w.lane16 = (unsigned)lane * 16u;
    stream_prime(w.ws, wstream, w.lds, w.tid16);

    u32x4 bh[2][8], bl[2][8];
    f32x4 acc[2][2];
    const float s0 = scale_for(1.f);
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = lab_input(p, (unsigned)(32 * s + 16 * (e >> 2) + 4 * g + (e & 3)));
        cut8(x, s0, bh[0][s], bl[0][s]);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[1][x][j] = lab_input(p, (unsigned)(16 * (14 + x) + 4 * g + j));
    block_sync();
    ring_prime(w);

    const unsigned lane_store = tile_native_lane_offset(wv, lane);
    auto section = [&](int sect) { return uniform_global_rw(save + (long)sect * 256 * Ppad + tile32 * (32L * 256)); };
    auto scale_of = [&](int layer, int what) { return sc[layer * h3::kScaleStride + what]; };
    auto amax_of = [&](float a) {
        a = fmaxf(a, shfl_xor(a, 16));
        return fmaxf(a, shfl_xor(a, 32));
    };
    using Relu = ReluEpiP<TRAIN>;
    auto make_relu = [&](int l, float s_in) {
        Relu e;
        e.os = inv_pow2(s_in) * scale_of(l, h3::kSwInv);
        e.s_next = 1.f;
        e.am = 0.f;
        e.bias = tables + 256 * l + 4 * g;
        e.save = TRAIN ? section(l) : nullptr;
        e.lane_store = lane_store;
        return e;
    };
    auto store_mask = [&](Relu& epi, int sect) {
        if constexpr (TRAIN) {
            unsigned* base = reinterpret_cast<unsigned*>(save + (long)kSections * 256 * Ppad);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            store_at(uniform_global_rw(base + ((long)sect * (Ppad / 16) + tile32 * 2 + (wv & 1)) * 128), (unsigned)lane * 8u,
                     u32x2{epi.words[0], epi.words[1]});
        }
    };

    // a layer: input buffer X (its tiles 14, 15 = slab 7 still to come from `pend`), output buffer X ^ 1; leaves its own last
    // pair pending
    auto trunk_layer = [&](auto x_tag, Relu& pend, Relu& cur, int pend_sect, int layer) __attribute__((always_inline)) {
        constexpr int X = decltype(x_tag)::value;
        auto operand = [&](auto s_tag, u32x4& xh, u32x4& xl) {
            constexpr int s = decltype(s_tag)::value;
            xh = bh[X][s]; xl = bl[X][s];
        };
        tile_pair<0, 8>(w, acc[0], operand, [&](auto sg) { epi_slot<Relu, 7, decltype(sg)::value, 18>(pend, acc[1], bh[X], bl[X]); });
        store_mask(pend, pend_sect);
        const float am = amax_of(pend.am);
        cur.s_next = scale_for(__builtin_fmaf(scale_of(layer, h3::kBoundA), am, scale_of(layer, h3::kBoundB)));
        static_for<7>([&](auto q_tag) {
            constexpr int Q = decltype(q_tag)::value + 1;          // pair Q under it: the epilogue of pair Q - 1
            tile_pair<8 * Q, 8>(w, acc[Q & 1], operand, [&](auto sg) {
                epi_slot<Relu, Q - 1, decltype(sg)::value, 24>(cur, acc[(Q - 1) & 1], bh[X ^ 1], bl[X ^ 1]);
            });
        });
    };

    Relu prev = make_relu(kLayers, 1.f);        // the "layer" in front of the chain: acc = the input itself, bias 0
    prev.os = 1.f;
    prev.s_next = s0;
#pragma unroll 1
    for (int it = 0; it < kLayers / 2; ++it) {
        const int l = 2 * it;
        Relu cur = make_relu(l, prev.s_next);
        trunk_layer(I<0>{}, prev, cur, l == 0 ? kLayers : l - 1, l);
        prev = cur;
        Relu cur2 = make_relu(l + 1, prev.s_next);
        trunk_layer(I<1>{}, prev, cur2, l, l + 1);
        prev = cur2;
    }
    epi_all<Relu, 7>(prev, acc[1], bh[0], bl[0]);
    store_mask(prev, kLayers - 1);
    unsigned sum = 0u;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c) sum ^= bh[0][s][c] + 3u * bl[0][s][c];
    if (!TRAIN || sum == 0x12345u) out[(long)blockIdx.x * kThreadsP + threadIdx.x] = sum;
}


// ---- kind 6: today's residency with GATE-COMPACTED training stores (round 6 lab: fewer HBM bytes per sample?) ----------------
// A layer's post-ReLU output is ~half zeros and its gate bits are stored anyway.  Here a wave tile's block of a section holds
// only the non-zero values, densely: per accumulator register (= one feature group of the 64 lanes) the ballot of `v > 0` is
// the gate word pair, a lane's slot is the running count + the number of set bits below it (v_mbcnt), and the store runs
// with exec = the ballot -- 4-byte stores into consecutive addresses.  The 128 ballots of a layer are the mask words
// (collected with v_writelane: word 2 b + half of ballot b = (tile, quarter, element)).  What it costs the epilogue is what
// this lab measures (tools/residency_lab.py --kinds h3,h3-compact; bytes under --pmc).
#ifndef SCNERF_SIMT_EMU_BUILD
struct CompactEpi {
    float os, s_next, am;
    const float* bias;
    global_bytes_rw save;      // this wave tile's block of the layer's section: the packed values from its start
    unsigned lane16;
    unsigned run;              // bytes written so far (wave-uniform)
    unsigned words[4];
    f32x4 bq;
    float v[4];
    unsigned hp;

    // (an epilogue that only ever runs its pending pair -- the "layer" in front of the chain -- starts here too)
    __device__ __forceinline__ void prime() { run = 0u; words[0] = words[1] = words[2] = words[3] = 0u; }

    template <int P, int PIECE, int SUB, int NS>
    __device__ __forceinline__ void sub(f32x16 (&acc)[2], h3::u32x4 (&oh)[NS], h3::u32x4 (&ol)[NS]) {
        constexpr int x = PIECE >> 2, q = PIECE & 3, T = 2 * P + x;
        constexpr int sl = 2 * T + (q >> 1), c0 = 2 * (q & 1);
        if constexpr (SUB == 0) {
            bq = *reinterpret_cast<const f32x4*>(bias + (4 * T + q) * 8);
        } else if constexpr (SUB <= 4) {
            constexpr int e = SUB - 1;
            constexpr int b = (4 * T + q) * 4 + e;               // ballot index within the layer: 0 .. 127
            v[e] = relu_raw(__builtin_fmaf(acc[x][4 * q + e], os, bq[e]));
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(v[e] > 0.f);
            const unsigned lo = (unsigned)mask, hi = (unsigned)(mask >> 32);
            const unsigned below = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
            const global_bytes_rw at = uniform_global_rw(save + run);
            asm volatile("s_mov_b64 exec, %0\n\tglobal_store_dword %1, %2, %3 nt\n\ts_mov_b64 exec, -1"
                         :: "s"(mask), "v"(below * 4u), "v"(v[e]), "s"(at) : "memory");
            run += 4u * (unsigned)__builtin_popcountll(mask);
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(words[(2 * b) >> 6]) : "s"(lo), "n"((2 * b) & 63));
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(words[(2 * b + 1) >> 6]) : "s"(hi), "n"((2 * b + 1) & 63));
        } else if constexpr (SUB == 5) {
            am = max3(am, v[0], v[1]);
            am = max3(am, v[2], v[3]);
        } else if constexpr (SUB == 6) {
            hp = pack_f16_scaled(v[0], v[1], s_next);
        } else if constexpr (SUB == 7) {
            oh[sl][c0] = hp;
            ol[sl][c0] = pack_f16(residual_f16<0>(v[0], s_next, hp), residual_f16<1>(v[1], s_next, hp));
        } else if constexpr (SUB == 8) {
            hp = pack_f16_scaled(v[2], v[3], s_next);
        } else if constexpr (SUB == 9) {
            oh[sl][c0 + 1] = hp;
            ol[sl][c0 + 1] = pack_f16(residual_f16<0>(v[2], s_next, hp), residual_f16<1>(v[3], s_next, hp));
        }
    }
};
#endif


// ---- kinds 7 .. 11: today's residency with the activation stores under another cache policy (round 6: do the 8 GB of stores of a
// training forward cost less energy -- the launch is power-bound -- when they by-pass / write through some level?).  The product
// stores with `nt` (__builtin_nontemporal_store); gfx950's store modifiers are sc0, sc1 (scope) and nt.
#ifndef SCNERF_SIMT_EMU_BUILD
template <int POLICY>
struct PolicyEpi : h3f::FwdEpi<true, 0> {
    using Base = h3f::FwdEpi<true, 0>;
    template <int P, int PIECE, int SUB, int NS>
    __device__ __forceinline__ void sub(f32x16 (&acc)[2], h3::u32x4 (&oh)[NS], h3::u32x4 (&ol)[NS]) {
        if constexpr (SUB == 10) {
            constexpr int q = PIECE & 3, T = 2 * P + (PIECE >> 2);
            const global_bytes_rw at = uniform_global_rw(this->save + (4 * T + q) * 1024);
            const f32x4 val = {this->v[0], this->v[1], this->v[2], this->v[3]};
            const unsigned off = pinned_here(this->lane16);
            // (s_nop 1: a store of more than 8 bytes must not be followed at once by a write of its data registers, and the
            //  hazard recogniser does not look inside an asm -- the first version of this lab lacked it and its sampled check
            //  did not notice the one value in a thousand that was wrong)
            if constexpr (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(off), "v"(val), "s"(at) : "memory");
            else if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" :: "v"(off), "v"(val), "s"(at) : "memory");
            else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt\n\ts_nop 1" :: "v"(off), "v"(val), "s"(at) : "memory");
            else if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" :: "v"(off), "v"(val), "s"(at) : "memory");
            else if constexpr (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" :: "v"(off), "v"(val), "s"(at) : "memory");
            else if constexpr (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, %2 sc1 nt\n\ts_nop 1" :: "v"(off), "v"(val), "s"(at) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, %2 sc0 nt\n\ts_nop 1" :: "v"(off), "v"(val), "s"(at) : "memory");
        } else {
            Base::template sub<P, PIECE, SUB, NS>(acc, oh, ol);
        }
    }
};
#endif

// ---- kind 0: today's residency (mlp_h3.h, mlp_fwd_h3_kernel.h's epilogue) ----------------------------------------------
constexpr unsigned kLds32 = h3::kStreamLds + kSections * 256 * 4;

template <bool TRAIN, class EPI = h3f::FwdEpi<TRAIN, 0>>
__global__ __launch_bounds__(kThreads, 1) void chain32_kernel(const short* __restrict__ wstream, const float* __restrict__ bias,
                                                              const float* __restrict__ sc, float* __restrict__ save_arg,
                                                              unsigned* __restrict__ out, long P) {
    using namespace scn::h3;
    float* const save = TRAIN ? save_arg : nullptr;
    const long Ppad = padded_samples(P);
    Wave w;
    w.lds = dynamic_lds<char>();
    float* const tables = reinterpret_cast<float*>(w.lds + kStreamLds);
    for (int i = threadIdx.x; i < kSections * 256; i += kThreads) tables[i] = i < kLayers * 256 ? bias[i] : 0.f;
    const int lane = lane_id(), m = lane & 31, h = lane >> 5;
    const long wave_tile = (long)blockIdx.x * 4 + uniform(wave_id());
    const unsigned p = (unsigned)(wave_tile * 32 + m);
    w.tid16 = threadIdx.x * 16u;
    w.lane16 = (unsigned)lane * 16u;
    stream_prime(w.ws, wstream, w.lds, w.tid16);

    u32x4 bh[2][16], bl[2][16];
    f32x16 acc[2][2];
    const float s0 = scale_for(1.f);
#pragma unroll
    for (int s = 0; s < 12; ++s) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = lab_input(p, (unsigned)(32 * (s >> 1) + 16 * (s & 1) + 8 * (e >> 2) + 4 * h + (e & 3)));
        cut8(x, s0, bh[0][s], bl[0][s]);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[1][x][r] = lab_input(p, (unsigned)(32 * (6 + x) + 8 * (r >> 2) + 4 * h + (r & 3)));
    block_sync();
    ring_prime(w);

    auto section = [&](int sect) { return uniform_global_rw(save + (long)sect * 256 * Ppad + wave_tile * (32L * 256)); };
    auto scale_of = [&](int layer, int what) { return sc[layer * kScaleStride + what]; };
    auto amax_of = [&](float a) { return fmaxf(a, shfl_xor(a, 32)); };
    using Relu = EPI;
    auto make_relu = [&](int l, float s_in) {
        Relu e;
        e.os = inv_pow2(s_in) * scale_of(l, kSwInv);
        e.s_next = 1.f;
        e.am = 0.f;
        e.bias = tables + 256 * l + 4 * h;
        e.save = TRAIN ? section(l) : nullptr;
        e.lane16 = w.lane16;
        e.prime();
        return e;
    };
    auto store_mask = [&](Relu& epi, int sect) {
        if constexpr (TRAIN) {
            unsigned* base = reinterpret_cast<unsigned*>(save + (long)kSections * 256 * Ppad);
            store_at(uniform_global_rw(base + ((long)sect * (Ppad / 32) + wave_tile) * 256), w.lane16,
                     u32x4{epi.words[0], epi.words[1], epi.words[2], epi.words[3]});
        }
    };
    auto trunk_layer = [&](auto x_tag, Relu& pend, Relu& cur, int pend_sect, int layer) __attribute__((always_inline)) {
        constexpr int X = decltype(x_tag)::value;
        auto operand = [&](auto s_tag, u32x4& xh, u32x4& xl) {
            constexpr int s = decltype(s_tag)::value;
            xh = bh[X][s]; xl = bl[X][s];
        };
        tile_pair<0, 16>(w, acc[0], operand, [&](auto sg) { epi_slot<Relu, 3, decltype(sg)::value, 9>(pend, acc[1], bh[X], bl[X]); });
        store_mask(pend, pend_sect);
        const float am = amax_of(pend.am);
        cur.s_next = scale_for(__builtin_fmaf(scale_of(layer, kBoundA), am, scale_of(layer, kBoundB)));
        tile_pair<16, 16>(w, acc[1], operand, [&](auto sg) { epi_slot<Relu, 0, decltype(sg)::value, 12>(cur, acc[0], bh[X ^ 1], bl[X ^ 1]); });
        tile_pair<32, 16>(w, acc[0], operand, [&](auto sg) { epi_slot<Relu, 1, decltype(sg)::value, 12>(cur, acc[1], bh[X ^ 1], bl[X ^ 1]); });
        tile_pair<48, 16>(w, acc[1], operand, [&](auto sg) { epi_slot<Relu, 2, decltype(sg)::value, 12>(cur, acc[0], bh[X ^ 1], bl[X ^ 1]); });
    };

    Relu prev = make_relu(kLayers, 1.f);
    prev.os = 1.f;
    prev.s_next = s0;
#pragma unroll 1
    for (int it = 0; it < kLayers / 2; ++it) {
        const int l = 2 * it;
        Relu cur = make_relu(l, prev.s_next);
        trunk_layer(I<0>{}, prev, cur, l == 0 ? kLayers : l - 1, l);
        prev = cur;
        Relu cur2 = make_relu(l + 1, prev.s_next);
        trunk_layer(I<1>{}, prev, cur2, l, l + 1);
        prev = cur2;
    }
    epi_all<Relu, 3>(prev, acc[1], bh[0], bl[0]);
    store_mask(prev, kLayers - 1);
    unsigned sum = 0u;
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c) sum ^= bh[0][s][c] + 3u * bl[0][s][c];
    if (!TRAIN || sum == 0x12345u) out[(long)blockIdx.x * kThreads + threadIdx.x] = sum;
}

template <class K>
static int launch_timed(K kernel, unsigned lds, int threads, const short* wstream, const float* bias, const float* sc, float* save,
                        unsigned* out, long long P, int reps, float* ms) {
    SCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid(scn_ceil_div(P, 128));
#ifndef SCNERF_SIMT_EMU_BUILD
    hipEvent_t e0, e1;
    SCN_HIP(hipEventCreate(&e0));
    SCN_HIP(hipEventCreate(&e1));
    for (int r = 0; r < reps; ++r) {
        SCN_HIP(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, nullptr, wstream, bias, sc, save, out, (long)P);
        SCN_HIP(hipEventRecord(e1, nullptr));
        SCN_HIP(hipEventSynchronize(e1));
        SCN_HIP(hipEventElapsedTime(&ms[r], e0, e1));
    }
    SCN_HIP(hipEventDestroy(e0));
    SCN_HIP(hipEventDestroy(e1));
#else
    (void)reps;
    hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, nullptr, wstream, bias, sc, save, out, (long)P);
    ms[0] = 0.f;
#endif
    return scn_launch_status();
}

}  // namespace lab_chain

// kind: 0 h3 train, 1 h3p train, 2 h3 inference, 3 h3p inference.  ms[reps] receives the launch times (HIP events on the
// null stream).  save: kSections x 256 x padded(P) floats + mask words; out: one word per thread.
extern "C" int residency_lab_run(int kind, const void* wstream, const void* bias, const void* sc, void* save, void* out,
                                 long long P, int reps, float* ms) {
    using namespace lab_chain;
    const short* ws = static_cast<const short*>(wstream);
    const float* b = static_cast<const float*>(bias);
    const float* s = static_cast<const float*>(sc);
    float* sv = static_cast<float*>(save);
    unsigned* o = static_cast<unsigned*>(out);
    switch (kind) {
        case 0: return launch_timed(chain32_kernel<true>, kLds32, scn::mlp::kThreads, ws, b, s, sv, o, P, reps, ms);
        case 1: return launch_timed(chain16_kernel<true>, kLdsP, scn::h3p::kThreadsP, ws, b, s, sv, o, P, reps, ms);
        case 2: return launch_timed(chain32_kernel<false>, kLds32, scn::mlp::kThreads, ws, b, s, sv, o, P, reps, ms);
        case 3: return launch_timed(chain16_kernel<false>, kLdsP, scn::h3p::kThreadsP, ws, b, s, sv, o, P, reps, ms);
#ifndef SCNERF_SIMT_EMU_BUILD
        case 6: return launch_timed(chain32_kernel<true, CompactEpi>, kLds32, scn::mlp::kThreads, ws, b, s, sv, o, P, reps, ms);
        case 7: return launch_timed(chain32_kernel<true, PolicyEpi<0>>, kLds32, scn::mlp::kThreads, ws, b, s, sv, o, P, reps, ms);
        case 8: return launch_timed(chain32_kernel<true, PolicyEpi<1>>, kLds32, scn::mlp::kThreads, ws, b, s, sv, o, P, reps, ms);
        case 9: return launch_timed(chain32_kernel<true, PolicyEpi<2>>, kLds32, scn::mlp::kThreads, ws, b, s, sv, o, P, reps, ms);
        case 10: return launch_timed(chain32_kernel<true, PolicyEpi<3>>, kLds32, scn::mlp::kThreads, ws, b, s, sv, o, P, reps, ms);
        case 11: return launch_timed(chain32_kernel<true, PolicyEpi<4>>, kLds32, scn::mlp::kThreads, ws, b, s, sv, o, P, reps, ms);
        case 12: return launch_timed(chain32_kernel<true, PolicyEpi<5>>, kLds32, scn::mlp::kThreads, ws, b, s, sv, o, P, reps, ms);
        case 13: return launch_timed(chain32_kernel<true, PolicyEpi<6>>, kLds32, scn::mlp::kThreads, ws, b, s, sv, o, P, reps, ms);
#endif
    }
    return SCN_EINVAL;
}
