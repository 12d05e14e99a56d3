// mlp_fwd.hip -- fused NeRF MLP forward for gfx950: positional encoding -> 8-layer ReLU
// trunk with skip -> density head + view-dependent colour head, one launch, activations
// never leave the register file (mlp_common.h explains the layout).
//
// Replaces, for the standard SCNeRF network (D=8, W=256, skips=[4], use_viewdirs, multires
// 10 / 4): run_network + Embedder + NeRF.forward
//   /root/reference NeRF/create_nerf.py:18-32, NeRF/run_nerf_helpers.py:24-72, :105-128.
//
// Per workgroup: 4 waves x 32 samples.  9 344 v_mfma_f32_32x32x2_f32 per wave
// (= 1 196 032 issued MAC-pairs per sample vs 593 408 x 2 algorithmic: 99.2 % useful).
// Training mode additionally stores the post-ReLU activations, the colour-head inputs and
// the encodings row-major for the dgrad / wgrad kernels.
#include <scn_wave.h>

#include "launch.h"
#include "mlp_common.h"
#include "scnerf_hip.h"

namespace {

using namespace scn;
using namespace scn::mlp;


template <int N>
__device__ __forceinline__ void relu_to_regs(const f32x16 (&acc)[N / 16], float (&dst)[N], bool relu) {
#pragma unroll
    for (int t = 0; t < N / 16; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[16 * t + r] = relu ? max_raw(acc[t][r], 0.f) : acc[t][r];
}

// encodings are saved in torch column order so the wgrad GEMM writes weight columns directly
template <int PD, int L, int NS>
__device__ __forceinline__ void store_pe(const float (&e)[NS], float* __restrict__ base, long p, int ld,
                                         int h, bool live) {
    if (!live) return;
    float* row = base + p * ld;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c0 = pe_col(L, s, 0, PD), c1 = pe_col(L, s, 1, PD);
        const int c = h ? c1 : c0;
        if (c >= 0) row[c] = e[s];
    }
    if (h == 1) {
        // zero the pad columns of the row (they are read, masked, by the wgrad GEMM)
        for (int c = PD + 2 * PD * L; c < ld; ++c) row[c] = 0.f;
    }
}

// Epilogue of a trunk layer, register by register (LastChunk issues it between the MFMAs of the layer's
// last chunk): activation -> B-operand register of the next layer, ReLU mask bit, bias of the next layer
// into the accumulator.  `lo` = 0 (ReLU) or -inf (the linear feature layer).
template <bool TRAIN>
struct FwdEpi {
    static constexpr int kValuPerMfma = 5;
    f32x16 (&acc)[8];
    float (&hreg)[128];
    const float* bias_next;      // lane-vector table of the layer that follows
    int h;
    float lo;
    unsigned (&bits)[4];
    unsigned ha, hb;
    f32x4 ba, bb;
    template <int P, int G, int J>
    __device__ __forceinline__ void slice() {
        constexpr int r = 4 * G + J;
        if constexpr (G == 0 && J == 0) { ha = 0u; hb = 0u; }
        if constexpr (J == 0) {
            ba = lane_vec(bias_next, 2 * P, G, h);
            bb = lane_vec(bias_next, 2 * P + 1, G, h);
        }
        {
            constexpr int t = 2 * P, i = 16 * t + r;
            const float v = max_raw(acc[t][r], lo);
            hreg[i] = v;
            if constexpr (TRAIN) ha = shift_in_positive(ha, v);
            acc[t][r] = ba[J];
        }
        {
            constexpr int t = 2 * P + 1, i = 16 * t + r;
            const float v = max_raw(acc[t][r], lo);
            hreg[i] = v;
            if constexpr (TRAIN) hb = shift_in_positive(hb, v);
            acc[t][r] = bb[J];
        }
        if constexpr (TRAIN && G == 3 && J == 3) bits[P] = (ha << 16) | hb;      // element i: word i >> 5, bit 31 - (i & 31)
    }
};

// TRAIN: also leave the activations / encodings / ReLU masks in `save` for the dgrad and wgrad kernels
template <int PD, bool TRAIN>
__global__ __launch_bounds__(kThreads, 1) void mlp_fwd_kernel(
    const float* __restrict__ pts, const float* __restrict__ viewdirs, int vd_stride, int samples_per_ray,
    const float* __restrict__ wpk, float* __restrict__ raw, float* __restrict__ save_arg, long P) {
    float* const save = TRAIN ? save_arg : nullptr;
    const int lane = lane_id();
    const int m = lane & 31, h = lane >> 5;
    const long wave_tile = (long)blockIdx.x * 4 + wave_id();
    const long p = wave_tile * kSamplesPerWave + m;
    const bool live = p < P;
    const long pc = live ? p : P - 1;
    const long Ppad = padded_samples(P);
    using V = Var<PD>;
    constexpr int ES = V::kES;

    WStream ws;
    ws.g = reinterpret_cast<const f32x4*>(wpk);
    stream_prime<8>(ws);   // first chunk of E0 (8 tiles x 16 steps)

    const float px = pts[pc * PD + 0], py = pts[pc * PD + 1], pz = pts[pc * PD + 2];
    const float pw = PD == 4 ? pts[pc * PD + (PD - 1)] : 0.f;

    float hreg[256 / 2];           // this lane's 128 of the 256 trunk features
    f32x16 acc[8];
    float sigma_part = 0.f;

    // layer 0 (peeled: nothing else is live while the 30 sincos of the encoding run).  The encoded
    // point is parked in LDS for the skip layer instead of staying live through layers 1..4.
    f32x4* park = reinterpret_cast<f32x4*>(dynamic_lds<float>() + kStreamBufs * kMaxChunkFwd) + threadIdx.x;
    {
        float e[ES];
        pe_slots<PD, 10, ES>(px, py, pz, pw, h, e);
        if (save) store_pe<PD, 10, ES>(e, save + (long)kSaveEpts * Ppad, pc, V::kEW, h, live);
#pragma unroll
        for (int g = 0; g < ES / 4; ++g) {
            f32x4 v = {e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3]};
            park[g * kThreads] = v;
        }
        init_bias<8>(acc, wpk + V::kFwdBias, h);
        mfma_part<ES, 8, 16, 8>(e, acc, ws);
        relu_to_regs<128>(acc, hreg, true);
        if (save) *reinterpret_cast<u32x4*>(mask_ptr<PD>(save, P, 0, wave_tile, lane)) = relu_bits<128>(hreg);
    }

    // trunk layers 1..7 and the (linear) feature layer as l == 8.  In training mode the output of
    // layer l-1 (the B operand of layer l's main part) is written to HBM chunk by chunk during layer l.
    // Each layer's epilogue (ReLU -> next B operands, mask bits, next layer's bias) is folded into its last
    // chunk (FwdEpi / LastChunk), so the accumulators arrive here already holding the bias of layer l.
    init_bias<8>(acc, wpk + V::kFwdBias + 256, h);
#pragma unroll 1
    for (int l = 1; l <= 8; ++l) {
        if (l == 5) {
            float e[ES];
#pragma unroll
            for (int g = 0; g < ES / 4; ++g) {
                const f32x4 v = park[g * kThreads];
                e[4 * g] = v[0]; e[4 * g + 1] = v[1]; e[4 * g + 2] = v[2]; e[4 * g + 3] = v[3];
            }
            mfma_part<ES, 8, 16, 8>(e, acc, ws);
        }
        unsigned bits[4] = {0u, 0u, 0u, 0u};
        // bias of layer l + 1 (l == 8: the accumulators are not used again; any valid table)
        FwdEpi<TRAIN> epi{acc, hreg, wpk + (l < 7 ? V::kFwdBias + 256 * (l + 1) : V::kFwdBiasF), h,
                   l < 8 ? 0.f : -__builtin_huge_valf(), bits, 0u, 0u, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        mfma_part_epi<128, 8, 16, 8>(hreg, acc, ws,       // every chunk that can follow is 8 x 16 B / thread
                                     TRAIN ? tile_ptr(save + (long)(kSaveAct + 256 * (l - 1)) * Ppad, wave_tile, 256, lane) : nullptr,
                                     epi);
        if (save && l < 8) {
            u32x4 m = {bits[0], bits[1], bits[2], bits[3]};
            *reinterpret_cast<u32x4*>(mask_ptr<PD>(save, P, l, wave_tile, lane)) = m;
        }
        if (l == 7) {
            // density head on the VALU: sigma = w_alpha . h8 + b  (half of the features per lane)
            const float* wa = wpk + V::kFwdAlphaW;          // lane-vector layout
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = lane_vec(wa, t, q, h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) sigma_part = fmaf(w[j], hreg[16 * t + 4 * q + j], sigma_part);
                }
        }
    }

    // colour head: views layer on [feature | encoded view direction]
    const long ray = pc / samples_per_ray;
    float ev[16];
    pe_slots<3, 4, 16>(viewdirs[ray * vd_stride + 0], viewdirs[ray * vd_stride + 1], viewdirs[ray * vd_stride + 2], 0.f, h, ev);
    if (save) store_pe<3, 4, 16>(ev, save + (long)kSaveEviews * Ppad, pc, 32, h, live);
    f32x16 accv[4];
    init_bias<4>(accv, wpk + V::kFwdBiasV, h);
    mfma_part<128, 4, 32, 4>(hreg, accv, ws,             // then VE: 4 tiles x 16 steps = 4 f4
                             TRAIN ? tile_ptr(save + (long)kSaveFeat * Ppad, wave_tile, 256, lane) : nullptr);
    mfma_part<16, 4, 16, 4>(ev, accv, ws);               // then RGB: 1 tile x 64 steps = 4 f4
    float hv[64];
    relu_to_regs<64>(accv, hv, true);
    if (save) *reinterpret_cast<u32x4*>(mask_ptr<PD>(save, P, 8, wave_tile, lane)) = relu_bits<64>(hv);

    f32x16 accc[1];
    init_bias<1>(accc, wpk + V::kFwdBiasRGB, h);
    mfma_part<64, 1, 64, 0>(hv, accc, ws, TRAIN ? tile_ptr(save + (long)kSaveHv * Ppad, wave_tile, 128, lane) : nullptr);

    const float sigma = sigma_part + shfl_xor(sigma_part, 32) + wpk[V::kFwdAlphaB];
    if (live && h == 0) {
        // rows 0,1,2 of the single tile are registers 0,1,2 of the h == 0 half
        f32x4 o = {accc[0][0], accc[0][1], accc[0][2], sigma};
        *reinterpret_cast<f32x4*>(raw + p * 4) = o;
    }
}

__global__ void gather_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                              float* __restrict__ dst, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = idx[i];
    dst[i] = j >= 0 ? src[j] : 0.f;
}

}  // namespace

extern "C" int scnerf_gather_f32(const float* src, const int* idx, float* dst, long long n, void* stream) {
    SCN_RETURN_IF(!src || !idx || !dst || n < 0, SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(gather_kernel, dim3(scn_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       src, idx, dst, (long)n);
    return scn_launch_status();
}

template <int PD>
static void layout_values(int* v) {
    using V = Var<PD>;
    const int vals[] = {V::kFwdStream, V::kFwdBias, V::kFwdBiasF, V::kFwdBiasV, V::kFwdBiasRGB, V::kFwdAlphaW,
                        V::kFwdAlphaB, V::kFwdTotal, V::kBwdStream, V::kBwdAlphaW, V::kBwdTotal, V::kSavePerSample,
                        kGradPerSample, kSaveFeat, kSaveHv, kSaveEpts, kSaveEviews, kGradDfeat, kGradDzv,
                        kMaskWordsPerSample, V::kNParams, V::kEW};
    for (int i = 0; i < 22; ++i) v[i] = vals[i];
}

extern "C" int scnerf_mlp_layout_info(int pt_dims, int* out, int n) {
    SCN_RETURN_IF(!out || n < 22 || (pt_dims != 3 && pt_dims != 4), SCN_EINVAL);
    if (pt_dims == 3) layout_values<3>(out); else layout_values<4>(out);
    return 0;
}

extern "C" long long scnerf_mlp_save_floats(int pt_dims, long long n_samples) {
    const int per = pt_dims == 4 ? Var<4>::kSavePerSample : Var<3>::kSavePerSample;
    return (long long)(per + kMaskWordsPerSample) * padded_samples(n_samples);
}

extern "C" long long scnerf_mlp_grad_floats(long long n_samples) {
    return (long long)kGradPerSample * padded_samples(n_samples);
}

template <int PD, bool TRAIN>
static int launch_fwd(const float* pts, const float* viewdirs, int vd_stride, int samples_per_ray,
                      const float* wpacked, float* raw, float* save, long long n_samples, hipStream_t st) {
    // weights (3 x 32 KB) + the parked encoding of the 256 threads
    const size_t lds = (size_t)(kStreamBufs * kMaxChunkFwd + Var<PD>::kES * kThreads) * sizeof(float);
    SCN_LDS_OPT_IN((mlp_fwd_kernel<PD, TRAIN>), lds);
    hipLaunchKernelGGL((mlp_fwd_kernel<PD, TRAIN>), dim3(scn_ceil_div(n_samples, kSamplesPerBlock)), dim3(kThreads), lds,
                       st, pts, viewdirs, vd_stride, samples_per_ray, wpacked, raw, save, (long)n_samples);
    return scn_launch_status();
}

extern "C" int scnerf_mlp_fwd(int pt_dims, const float* pts, const float* viewdirs, int vd_stride,
                              int samples_per_ray, const float* wpacked, float* raw, float* save,
                              long long n_samples, void* stream) {
    SCN_RETURN_IF(!pts || !viewdirs || !wpacked || !raw || samples_per_ray < 1 || vd_stride < 3 || n_samples < 0, SCN_EINVAL);
    SCN_RETURN_IF(pt_dims != 3 && pt_dims != 4, SCN_EINVAL);
    if (n_samples == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (pt_dims == 3)
        return save ? launch_fwd<3, true>(pts, viewdirs, vd_stride, samples_per_ray, wpacked, raw, save, n_samples, st)
                    : launch_fwd<3, false>(pts, viewdirs, vd_stride, samples_per_ray, wpacked, raw, save, n_samples, st);
    return save ? launch_fwd<4, true>(pts, viewdirs, vd_stride, samples_per_ray, wpacked, raw, save, n_samples, st)
                : launch_fwd<4, false>(pts, viewdirs, vd_stride, samples_per_ray, wpacked, raw, save, n_samples, st);
}
