// mlp_bwd_h3_kernel.h -- the resident-arithmetic data-gradient kernel (see mlp_bwd_h3.hip) and its launch template; included
// by one translation unit per network variant (mlp_bwd_h3_pd3.hip, mlp_bwd_h3_pd4.hip), compiled side by side.
#pragma once
#include <scn_wave.h>

#include "launch.h"
#include "mlp_bwd_h3_api.h"
#include "mlp_h3.h"

namespace scn {
namespace h3b {

using namespace scn::mlp;
using namespace scn::h3;


// Gradient of the positional encoding held in slot layout (mlp_common.h pe_slots): this lane half's contributions
// (as mlp_bwd.hip's pe_backward).
template <int PD, int L, int NS>
__device__ __forceinline__ void pe_backward(float x, float y, float z, float w, int h, const float (&de)[NS], float* d_a, float* d_b) {
    const float a = h ? y : x;
    float freq = 1.f;
    if constexpr (PD == 3) {
        float ga = de[3 * L];
        float gz = h ? 0.f : de[3 * L + 1];
#pragma unroll
        for (int f = 0; f < L; ++f) {
            float s0, c0, s1, c1;
            sincos(a * freq, &s0, &c0);
            sincos(z * freq, &s1, &c1);
            ga += freq * (c0 * de[3 * f] - s0 * de[3 * f + 1]);
            gz += freq * ((h ? -s1 : c1) * de[3 * f + 2]);
            freq *= 2.f;
        }
        *d_a = ga;
        *d_b = gz;
    } else {
        const float b = h ? w : z;
        float ga = de[4 * L], gb = de[4 * L + 1];
#pragma unroll
        for (int f = 0; f < L; ++f) {
            float s0, c0, s1, c1;
            sincos(a * freq, &s0, &c0);
            sincos(b * freq, &s1, &c1);
            ga += freq * (c0 * de[4 * f] - s0 * de[4 * f + 1]);
            gb += freq * (c1 * de[4 * f + 2] - s1 * de[4 * f + 3]);
            freq *= 2.f;
        }
        *d_a = ga;
        *d_b = gb;
    }
}

struct NoRank1 {};
struct Rank1 {
    const float* alpha;        // the density head's weights, LDS lane-vector table + 4 h
    float dsigma;
    f32x4 wqs[2];              // read from LDS one piece ahead (piece parity), as mlp_fwd_h3.hip's bias
};
struct NoGate {};
struct Gate { u32x4 gate; };   // ReLU bits of the layer below: element 16 t + r <-> word t >> 1, bit 31 - (16 (t & 1) + r)

// KIND 0: ReLU gate;  1: gate + the density head's rank-1 term (feature_linear^T);  2: linear (d feature)
template <int KIND>
struct BwdEpi : std::conditional_t<KIND == 1, Rank1, NoRank1>, std::conditional_t<KIND != 2, Gate, NoGate> {
    float os, s_next, am;
    global_bytes_rw save;      // this wave tile's block of the gradient section the result goes to
    unsigned lane16;
    float v[4];
    unsigned hp;

    template <int P, int PIECE, int SUB, int NS>
    __device__ __forceinline__ void sub(f32x16 (&acc)[2], u32x4 (&oh)[NS], u32x4 (&ol)[NS]) {
        constexpr int x = PIECE >> 2, q = PIECE & 3, T = 2 * P + x;
        constexpr int sl = 2 * T + (q >> 1), c0 = 2 * (q & 1);
        static_assert(sl < NS, "operand buffer too small for this tile");
        if constexpr (lab::kNoEpilogue) {
            if constexpr (SUB == 1) {
                oh[sl][c0] = __float_as_uint(acc[x][4 * q]) & 0x3bff3bffu; oh[sl][c0 + 1] = __float_as_uint(acc[x][4 * q + 1]) & 0x3bff3bffu;
                ol[sl][c0] = __float_as_uint(acc[x][4 * q + 2]) & 0x3bff3bffu; ol[sl][c0 + 1] = __float_as_uint(acc[x][4 * q + 3]) & 0x3bff3bffu;
            }
        } else if constexpr (SUB == 0) {
        } else if constexpr (SUB <= 4) {
            constexpr int e = SUB - 1;
            float d = acc[x][4 * q + e] * os;
            if constexpr (KIND == 1) d = __builtin_fmaf(this->wqs[PIECE & 1][e], this->dsigma, d);
            if constexpr (KIND != 2) {
                constexpr int bit = 31 - (16 * (T & 1) + 4 * q + e);
                v[e] = keep_if_bit<bit>(d, this->gate[T >> 1]);
            } else {
                v[e] = d;
            }
        } else if constexpr (SUB == 5) {
            am = max3_abs(am, v[0], v[1]);
            am = max3_abs(am, v[2], v[3]);
        } else if constexpr (SUB == 6) {
            hp = pack_f16_scaled(v[0], v[1], s_next);
            if constexpr (KIND == 1) {
                constexpr int NT = PIECE < 7 ? 4 * (2 * P + ((PIECE + 1) >> 2)) + ((PIECE + 1) & 3) : (P < 3 ? 4 * (2 * P + 2) : 0);
                this->wqs[(PIECE + 1) & 1] = *reinterpret_cast<const f32x4*>(this->alpha + NT * 8);
            }
        } else if constexpr (SUB == 7) {
            oh[sl][c0] = hp;
            ol[sl][c0] = pack_f16(residual_f16<0>(v[0], s_next, hp), residual_f16<1>(v[1], s_next, hp));
        } else if constexpr (SUB == 8) {
            hp = pack_f16_scaled(v[2], v[3], s_next);
        } else if constexpr (SUB == 9) {
            oh[sl][c0 + 1] = hp;
            ol[sl][c0 + 1] = pack_f16(residual_f16<0>(v[2], s_next, hp), residual_f16<1>(v[3], s_next, hp));
        } else if constexpr (SUB == 10) {
            if constexpr (!lab::kNoStore)
                store_written_through_at(uniform_global_rw(save + (4 * T + q) * 1024), pinned_here(lane16), f32x4{v[0], v[1], v[2], v[3]});
        }
    }
};


template <int PD>
__host__ __device__ constexpr unsigned bwd_lds_bytes() {
    return (unsigned)(kStreamLds + 256 * 4 + Var<PD>::kES * kThreads * 4);
}

template <int PD>
__global__ __launch_bounds__(kThreads, 1) void mlp_bwd_h3_kernel(
    const float* __restrict__ d_raw, const float* __restrict__ pts, const float* __restrict__ viewdirs, int vd_stride,
    int samples_per_ray, const float* __restrict__ wbk, const short* __restrict__ wh3, const float* __restrict__ sc,
    const float* __restrict__ save, float* __restrict__ grads, float* __restrict__ d_pts, float* __restrict__ d_views, long P,
    ChunkMaxima cm) {
    using V = Var<PD>;
    constexpr int ES = V::kES, ET = V::kET;        // encoded-point slots; 32-column tiles of the d-encoding parts
    claim_whole_register_file();
    const int lane = lane_id();
    const int m = lane & 31, h = lane >> 5;
    const long wave_tile = (long)blockIdx.x * 4 + uniform(wave_id());      // (a scalar: what derives from it -- section bases, chunk index -- is SALU work)
    const long p = wave_tile * kSamplesPerWave + m;
    const bool live = p < P;
    const long pc = live ? p : P - 1;
    const long Ppad = padded_samples(P);

    Wave w;
    w.lds = dynamic_lds<char>();
    w.tid16 = threadIdx.x * 16u;
    w.lane16 = (unsigned)lane * 16u;
    float* const alpha_tab = reinterpret_cast<float*>(w.lds + kStreamLds);
    f32x4* const park = reinterpret_cast<f32x4*>(w.lds + kStreamLds + 256 * 4) + threadIdx.x;      // d encoded point
    stream_prime(w.ws, wh3, w.lds, w.tid16);
    alpha_tab[threadIdx.x] = wbk[V::kBwdAlphaW + threadIdx.x];

    // dead lanes (p >= P) must contribute exact zeros: their d_raw is forced to 0
    f32x4 dr = *reinterpret_cast<const f32x4*>(d_raw + pc * 4);
    if (!live) dr = f32x4{0.f, 0.f, 0.f, 0.f};
    const float dsigma = dr[3];

    u32x4 bh[2][16], bl[2][16];
    f32x16 acc[2][2];
    auto section = [&](int offset, int width) {
        return uniform_global_rw(grads + (long)offset * Ppad + wave_tile * (32L * width));
    };
    auto load_gate = [&](int sect) {
        const unsigned* base = reinterpret_cast<const unsigned*>(save + (long)V::kSavePerSample * Ppad);
        return load_at<u32x4>(uniform_global(base + ((long)sect * (Ppad / 32) + wave_tile) * 256), w.lane16);
    };
    auto scale_of = [&](int layer, int what) { return sc[layer * kScaleStride + what]; };
    auto amax_of = [&](float a) { return fmaxf(a, shfl_xor(a, 32)); };
    // (which chunk: once per wave, as a scalar -- inside the lambda it was a 64-bit division, ~100 instructions, per layer)
    const int chunk_of_tile = cm.amax ? uniform((int)((unsigned)(wave_tile * kSamplesPerWave) / (unsigned)cm.chunk)) : 0;
    auto note_chunk_max = [&](int job, float v) __attribute__((always_inline)) {
        if (cm.amax == nullptr) return;
        v = fmaxf(v, shfl_xor(v, 16)); v = fmaxf(v, shfl_xor(v, 8)); v = fmaxf(v, shfl_xor(v, 4));
        v = fmaxf(v, shfl_xor(v, 2)); v = fmaxf(v, shfl_xor(v, 1));
        if (lane_id() == 0) atomic_max_nonneg(cm.amax + (long)job * cm.n_chunks + chunk_of_tile, v);
    };
    using GateEpi = BwdEpi<0>;
    auto make_gate = [&](int layer, float s_in, int mask_sect, int out_offset, int out_width) {
        GateEpi e;
        e.os = inv_pow2(s_in) * scale_of(layer, kSwInv);
        e.s_next = 1.f;
        e.am = 0.f;
        e.save = section(out_offset, out_width);
        e.lane16 = w.lane16;
        e.gate = load_gate(mask_sect);
        return e;
    };

    // ---- rgb_linear^T: d hv = W_rgb^T d rgb (contraction over the three channels: one K slab, lane half 0) ----
    const float am_rgb = fmaxf(fabsf(dr[0]), fmaxf(fabsf(dr[1]), fabsf(dr[2])));
    const float s_rgb = scale_for(am_rgb);
    u32x4 rh, rl;
    {
        const float x8[8] = {h ? 0.f : dr[0], h ? 0.f : dr[1], h ? 0.f : dr[2], 0.f, 0.f, 0.f, 0.f, 0.f};
        cut8(x8, s_rgb, rh, rl);
    }
    GateEpi epiv = make_gate(kLayerRgb, s_rgb, 8, kGradDzv, 128);
    epiv.s_next = scale_for(scale_of(kLayerRgb, kBoundAT) * am_rgb);
    block_sync();                       // the first chunk and the table are in LDS
    ring_prime(w);
    {
        auto operand = [&](auto, u32x4& xh, u32x4& xl) { xh = rh; xl = rl; };
        tile_pair<0, 1>(w, acc[0], operand, NoFill{});
        tile_pair<1, 1>(w, acc[1], operand, NoFill{});
        epi_all<GateEpi, 0>(epiv, acc[0], bh[1], bl[1]);
        epi_all<GateEpi, 1>(epiv, acc[1], bh[1], bl[1]);
    }
    const float am_v = amax_of(epiv.am);
    note_chunk_max(8, am_v);            // dZ of the views layer: the A operand of its weight-gradient GEMMs

    // ---- views layer^T, encoded-direction rows first: d ev = (W_v^T)[256 ..] dZ_v -> d viewdirs ----
    auto hv_operand = [&](auto s_tag, u32x4& xh, u32x4& xl) {
        constexpr int s = decltype(s_tag)::value;
        xh = bh[1][s]; xl = bl[1][s];
    };
    {
        f32x16 acce[2];
        tile_single<2, 4>(w, acce, hv_operand, NoFill{});
        const float os = inv_pow2(epiv.s_next) * scale_of(kLayerViews, kSwInv);
        float dev[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dev[r] = (acce[0][r] + acce[1][r]) * os;
        const long ray = (long)((unsigned)pc / (unsigned)samples_per_ray);       // (sample indices fit 31 bits)
        // max(1, |direction|) bounds every column of the encoded direction
        note_chunk_max(11, fmaxf(fmaxf(1.f, fabsf(viewdirs[ray * vd_stride + 0])),
                                 fmaxf(fabsf(viewdirs[ray * vd_stride + 1]), fabsf(viewdirs[ray * vd_stride + 2]))));
        float gxy, gz;
        pe_backward<3, 4, 16>(viewdirs[ray * vd_stride + 0], viewdirs[ray * vd_stride + 1], viewdirs[ray * vd_stride + 2], 0.f, h, dev, &gxy, &gz);
        const float oxy = shfl_xor(gxy, 32), oz = shfl_xor(gz, 32);
        if (live && h == 0) {
            d_views[p * 3 + 0] = gxy;
            d_views[p * 3 + 1] = oxy;
            d_views[p * 3 + 2] = gz + oz;
        }
    }
    // ---- d feature = (W_v^T)[.. 256] dZ_v: four pairs of 8 slabs, a pair's epilogue under the next pair ----
    using LinEpi = BwdEpi<2>;
    LinEpi epif;
    epif.os = inv_pow2(epiv.s_next) * scale_of(kLayerViews, kSwInv);
    epif.s_next = scale_for(scale_of(kLayerViews, kBoundAT) * am_v);
    epif.am = 0.f;
    epif.save = section(kGradDfeat, 256);
    epif.lane16 = w.lane16;
    tile_pair<6, 8>(w, acc[0], hv_operand, NoFill{});
    tile_pair<14, 8>(w, acc[1], hv_operand, [&](auto sg) { epi_slot<LinEpi, 0, decltype(sg)::value, 6>(epif, acc[0], bh[0], bl[0]); });
    tile_pair<22, 8>(w, acc[0], hv_operand, [&](auto sg) { epi_slot<LinEpi, 1, decltype(sg)::value, 6>(epif, acc[1], bh[0], bl[0]); });
    tile_pair<30, 8>(w, acc[1], hv_operand, [&](auto sg) { epi_slot<LinEpi, 2, decltype(sg)::value, 6>(epif, acc[0], bh[0], bl[0]); });

    // ---- a 256 -> 256 transposed layer: operand buffer X (its tiles 6, 7 still to come from `pend`), result -> X ^ 1;
    // leaves its own last pair pending.  Every such layer starts at stream unit 6 modulo 8.
    auto trunk_layer = [&](auto x_tag, auto& pend, auto& cur, float bound_a, float bound_extra, int job)
                           __attribute__((always_inline)) {        // (not inlined, the operand arrays go to scratch)
        constexpr int X = decltype(x_tag)::value;
        auto operand = [&](auto s_tag, u32x4& xh, u32x4& xl) {
            constexpr int s = decltype(s_tag)::value;
            xh = bh[X][s]; xl = bl[X][s];
        };
        using Pend = std::remove_reference_t<decltype(pend)>;
        using Cur = std::remove_reference_t<decltype(cur)>;
        tile_pair<6, 16>(w, acc[0], operand, [&](auto sg) { epi_slot<Pend, 3, decltype(sg)::value, 9>(pend, acc[1], bh[X], bl[X]); });
        const float am_in = amax_of(pend.am);              // the layer's operand (dZ of weight-gradient job `job`) is complete
        note_chunk_max(job, am_in);
        cur.s_next = scale_for(__builtin_fmaf(bound_a, am_in, bound_extra));
        tile_pair<22, 16>(w, acc[1], operand, [&](auto sg) { epi_slot<Cur, 0, decltype(sg)::value, 12>(cur, acc[0], bh[X ^ 1], bl[X ^ 1]); });
        tile_pair<38, 16>(w, acc[0], operand, [&](auto sg) { epi_slot<Cur, 1, decltype(sg)::value, 12>(cur, acc[1], bh[X ^ 1], bl[X ^ 1]); });
        tile_pair<54, 16>(w, acc[1], operand, [&](auto sg) { epi_slot<Cur, 2, decltype(sg)::value, 12>(cur, acc[0], bh[X ^ 1], bl[X ^ 1]); });
    };

    // ---- feature_linear^T + alpha_linear^T: d h8 = W_f^T d feature + w_alpha d sigma, gated by layer 7's ReLU ----
    BwdEpi<1> epi7;
    epi7.os = inv_pow2(epif.s_next) * scale_of(kLayerFeat, kSwInv);
    epi7.s_next = 1.f; epi7.am = 0.f;
    epi7.save = section(kGradDz + 7 * 256, 256);
    epi7.lane16 = w.lane16;
    epi7.gate = load_gate(7);
    epi7.alpha = alpha_tab + 4 * h;
    epi7.dsigma = dsigma;
    epi7.wqs[0] = *reinterpret_cast<const f32x4*>(epi7.alpha);
    trunk_layer(I<0>{}, epif, epi7, scale_of(kLayerFeat, kBoundAT), scale_of(kLayerAlpha, kBoundAT) * fabsf(dsigma), 7);
    // ---- trunk layers 7^T, 6^T: dZ_{l-1} = gate_{l-1}(W_l^T dZ_l) ----
    GateEpi prev = make_gate(7, epi7.s_next, 6, kGradDz + 6 * 256, 256);
    trunk_layer(I<1>{}, epi7, prev, scale_of(7, kBoundAT), 0.f, 6);
    {
        GateEpi cur = make_gate(6, prev.s_next, 5, kGradDz + 5 * 256, 256);
        trunk_layer(I<0>{}, prev, cur, scale_of(6, kBoundAT), 0.f, 5);
        prev = cur;
    }
    // ---- the skip layer: its encoded-point columns first (d encoded point, parked in LDS), then its h columns ----
    {
        auto operand = [&](auto s_tag, u32x4& xh, u32x4& xl) {
            constexpr int s = decltype(s_tag)::value;
            xh = bh[1][s]; xl = bl[1][s];
        };
        tile_pair<6, 16>(w, acc[0], operand, [&](auto sg) { epi_slot<GateEpi, 3, decltype(sg)::value, 9>(prev, acc[1], bh[1], bl[1]); });
        const float os5 = inv_pow2(prev.s_next) * scale_of(5, kSwInv);
        auto park_pair = [&](auto t0_tag, f32x16 (&a)[2]) {
            constexpr int T0 = decltype(t0_tag)::value;
#pragma unroll
            for (int x = 0; x < 2; ++x)
                if (16 * (T0 + x) < ES) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        park[(4 * (T0 + x) + q) * kThreads] = f32x4{a[x][4 * q] * os5, a[x][4 * q + 1] * os5, a[x][4 * q + 2] * os5, a[x][4 * q + 3] * os5};
                }
        };
        park_pair(I<0>{}, acc[0]);
        if constexpr (ET == 4) {
            tile_pair<22, 16>(w, acc[1], operand, NoFill{});
            park_pair(I<2>{}, acc[1]);
        }
        GateEpi cur = make_gate(5, prev.s_next, 4, kGradDz + 4 * 256, 256);
        const float am5 = amax_of(prev.am);
        note_chunk_max(4, am5);
        cur.s_next = scale_for(scale_of(5, kBoundAT) * am5);
        constexpr int U0 = 6;          // (the encoded-point part is 16 or 32 units: the phase stays)
        tile_pair<U0, 16>(w, acc[0], operand, NoFill{});
        tile_pair<U0 + 16, 16>(w, acc[1], operand, [&](auto sg) { epi_slot<GateEpi, 0, decltype(sg)::value, 12>(cur, acc[0], bh[0], bl[0]); });
        tile_pair<U0 + 32, 16>(w, acc[0], operand, [&](auto sg) { epi_slot<GateEpi, 1, decltype(sg)::value, 12>(cur, acc[1], bh[0], bl[0]); });
        tile_pair<U0 + 48, 16>(w, acc[1], operand, [&](auto sg) { epi_slot<GateEpi, 2, decltype(sg)::value, 12>(cur, acc[0], bh[0], bl[0]); });
        prev = cur;
    }
    // ---- trunk layers 4^T .. 1^T ----
#pragma unroll 1
    for (int l = 4; l >= 2; l -= 2) {
        GateEpi cur = make_gate(l, prev.s_next, l - 1, kGradDz + (l - 1) * 256, 256);
        trunk_layer(I<0>{}, prev, cur, scale_of(l, kBoundAT), 0.f, l - 1);
        GateEpi cur2 = make_gate(l - 1, cur.s_next, l - 2, kGradDz + (l - 2) * 256, 256);
        trunk_layer(I<1>{}, cur, cur2, scale_of(l - 1, kBoundAT), 0.f, l - 2);
        prev = cur2;
    }
    // ---- layer 0^T: d encoded point += W_0^T dZ_0, then the encoding's own gradient -> d pts ----
    {
        auto operand = [&](auto s_tag, u32x4& xh, u32x4& xl) {
            constexpr int s = decltype(s_tag)::value;
            xh = bh[0][s]; xl = bl[0][s];
        };
        float de[ES];
        tile_pair<6, 16>(w, acc[0], operand, [&](auto sg) { epi_slot<GateEpi, 3, decltype(sg)::value, 9>(prev, acc[1], bh[0], bl[0]); });
        const float os0 = inv_pow2(prev.s_next) * scale_of(0, kSwInv);
        note_chunk_max(9, amax_of(prev.am));        // dZ of layer 0
        auto add_pair = [&](auto t0_tag, f32x16 (&a)[2]) {
            constexpr int T0 = decltype(t0_tag)::value;
#pragma unroll
            for (int x = 0; x < 2; ++x)
                if (16 * (T0 + x) < ES) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 pk = park[(4 * (T0 + x) + q) * kThreads];
#pragma unroll
                        for (int j = 0; j < 4; ++j) de[16 * (T0 + x) + 4 * q + j] = __builtin_fmaf(a[x][4 * q + j], os0, pk[j]);
                    }
                }
        };
        add_pair(I<0>{}, acc[0]);
        if constexpr (ET == 4) {
            tile_pair<22, 16>(w, acc[1], operand, NoFill{});
            add_pair(I<2>{}, acc[1]);
        }
        float ga, gb;
        {
            // max(1, |point|) bounds every column of the encoded point (the point itself, sines and cosines)
            float am_e = fmaxf(fmaxf(1.f, fabsf(pts[pc * PD + 0])), fmaxf(fabsf(pts[pc * PD + 1]), fabsf(pts[pc * PD + 2])));
            if constexpr (PD == 4) am_e = fmaxf(am_e, fabsf(pts[pc * PD + 3]));
            note_chunk_max(10, am_e);
        }
        pe_backward<PD, 10, ES>(pts[pc * PD + 0], pts[pc * PD + 1], pts[pc * PD + 2], PD == 4 ? pts[pc * PD + (PD - 1)] : 0.f, h, de, &ga, &gb);
        const float oa = shfl_xor(ga, 32), ob = shfl_xor(gb, 32);
        if (live && h == 0) {
            d_pts[p * PD + 0] = ga;
            d_pts[p * PD + 1] = oa;
            if constexpr (PD == 3) {
                d_pts[p * PD + 2] = gb + ob;
            } else {
                d_pts[p * PD + 2] = gb;
                d_pts[p * PD + (PD - 1)] = ob;
            }
        }
    }
}


template <int PD>
inline int launch_bwd_h3(const float* d_raw, const float* pts, const float* viewdirs, int vd_stride, int samples_per_ray,
                         const float* wpacked_bwd, const short* stream_bwd, const float* scales, const float* save,
                         float* grads, float* d_pts, float* d_views, long long n_samples, ChunkMaxima cm, hipStream_t st) {
    constexpr unsigned lds = bwd_lds_bytes<PD>();
    SCN_LDS_OPT_IN((mlp_bwd_h3_kernel<PD>), lds);
    hipLaunchKernelGGL((mlp_bwd_h3_kernel<PD>), dim3(scn_ceil_div(n_samples, kSamplesPerBlock)), dim3(kThreads), lds, st,
                       d_raw, pts, viewdirs, vd_stride, samples_per_ray, wpacked_bwd, stream_bwd, scales, save, grads, d_pts,
                       d_views, (long)n_samples, cm);
    return scn_launch_status();
}


}  // namespace h3b
}  // namespace scn
