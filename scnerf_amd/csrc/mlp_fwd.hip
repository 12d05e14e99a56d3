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
#include "ray_stage.h"
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

// The coarse stage of render_rays as ONE launch (reference NeRF/render.py:235-262): stratified depths and points in
// the network kernel's prologue, alpha compositing in its epilogue.  64 samples per ray = two wave tiles, so a
// workgroup holds two whole rays: their raw outputs and depths meet in LDS and one wave per ray runs the same
// compositing code as composite_fwd_kernel (ray_stage.h), bit for bit.
struct CoarseStage {
    const float* rays; int ray_stride; int n_rays;
    const float* t_vals; const float* t_rand; int lindisp;     // t_rand [n_rays, 64] or nullptr (no jitter)
    float* z; float* pts;                                      // out: [n_rays, 64], [n_rays, 64, 3]
    const float* noise; int white_bkgd;                        // density noise [n_rays, 64] or nullptr
    float* rgb; float* disp; float* acc; float* depth; float* weights;   // out: compositing
};
constexpr int kCoarseSamples = 64;

// TRAIN: also leave the activations / encodings / ReLU masks in `save` for the dgrad and wgrad kernels
// COARSE: points come from `cs` (PD == 3, 64 samples per ray), `pts` is unused
template <int PD, bool TRAIN, bool COARSE = false>
__global__ __launch_bounds__(kThreads, 1) void mlp_fwd_kernel(
    const float* __restrict__ pts, const float* __restrict__ viewdirs, int vd_stride, int samples_per_ray,
    const float* __restrict__ wpk, float* __restrict__ raw, float* __restrict__ save_arg, long P, CoarseStage cs) {
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
    // stream order: E0, layers 1-4, skip part, layer 5 main, 6, 7, feature, views, view encoding, rgb
    ws.g = reinterpret_cast<const f32x4*>(wpk);
    stream_prime<8>(ws);   // first chunk of E0 (8 tiles x 16 steps)

    float px = 0.f, py = 0.f, pz = 0.f, pw = 0.f;
    auto coarse_depth_of_sample = [&]() {          // this lane's stratified depth (COARSE only; cheap to redo)
        const float* r = cs.rays + (pc >> 6) * cs.ray_stride;
        return ray::coarse_z(r[6], r[7], cs.t_vals, (int)(pc & 63), kCoarseSamples, cs.lindisp, cs.t_rand != nullptr,
                             cs.t_rand ? cs.t_rand[pc] : 0.f);
    };
    if constexpr (COARSE) {
        static_assert(PD == 3, "the coarse stage samples 3-D points");
        const float* r = cs.rays + (pc >> 6) * cs.ray_stride;
        const float z = coarse_depth_of_sample();
        px = r[0] + r[3] * z;
        py = r[1] + r[4] * z;
        pz = r[2] + r[5] * z;
        if (live && h == 0) {
            cs.z[p] = z;
            cs.pts[p * 3 + 0] = px; cs.pts[p * 3 + 1] = py; cs.pts[p * 3 + 2] = pz;
        }
    } else {
        px = pts[pc * PD + 0]; py = pts[pc * PD + 1]; pz = pts[pc * PD + 2];
        if constexpr (PD == 4) pw = pts[pc * PD + (PD - 1)];
    }

    float hreg[256 / 2];           // this lane's 128 of the 256 trunk features
    f32x16 acc[8];
    float sigma_part = 0.f;

    // layer 0 (peeled: nothing else is live while the 30 sincos of the encoding run).  The encoded
    // point is parked in LDS for the skip layer instead of staying live through layers 1..4.
    f32x4* park = reinterpret_cast<f32x4*>(dynamic_lds<float>() + kStreamBufs * kMaxChunkFwd) + threadIdx.x;
    auto density_head = [&]() {
        // sigma = w_alpha . h8 + b on the VALU (half of the features per lane)
        const float* wa = wpk + V::kFwdAlphaW;          // lane-vector layout
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = lane_vec(wa, t, q, h);
#pragma unroll
                for (int j = 0; j < 4; ++j) sigma_part = fmaf(w[j], hreg[16 * t + 4 * q + j], sigma_part);
            }
    };
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
        if (l == 7) density_head();
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
    // rows 0,1,2 of the single tile are registers 0,1,2 of the h == 0 half
    const f32x4 o = {accc[0][0], accc[0][1], accc[0][2], sigma};
    if (live && h == 0) *reinterpret_cast<f32x4*>(raw + p * 4) = o;
    if constexpr (COARSE) {
        // the parked-encoding area of the LDS is free (last read in layer 5, many chunk barriers ago)
        float* sraw = dynamic_lds<float>() + kStreamBufs * kMaxChunkFwd;      // [128 samples][4]
        float* sz = sraw + kSamplesPerBlock * 4;                              // [128]
        const int local = wave_id() * kSamplesPerWave + m;
        if (h == 0) {
            *reinterpret_cast<f32x4*>(sraw + local * 4) = o;
            sz[local] = coarse_depth_of_sample();
        }
        block_sync();
        if (wave_id() < kSamplesPerBlock / kCoarseSamples) {                  // one wave per ray, lane = sample
            const int slot = wave_id();
            long ray = (long)blockIdx.x * (kSamplesPerBlock / kCoarseSamples) + slot;
            const bool ray_live = ray < cs.n_rays;
            if (!ray_live) ray = cs.n_rays - 1;
            const float norm = ray::ray_norm(cs.rays + ray * cs.ray_stride + 3);
            auto fetch = [&](int i, f32x4* rw, float* zi) {
                *rw = *reinterpret_cast<const f32x4*>(sraw + (slot * kCoarseSamples + i) * 4);
                *zi = sz[slot * kCoarseSamples + i];
            };
            ray::composite_ray(fetch, kCoarseSamples, norm, cs.noise ? cs.noise + ray * kCoarseSamples : nullptr,
                               cs.white_bkgd, ray_live, lane, cs.rgb + ray * 3, cs.disp + ray, cs.acc + ray,
                               cs.depth ? cs.depth + ray : nullptr, cs.weights ? cs.weights + ray * kCoarseSamples : nullptr);
        }
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
    SCN_LDS_OPT_IN((mlp_fwd_kernel<PD, TRAIN, false>), lds);
    hipLaunchKernelGGL((mlp_fwd_kernel<PD, TRAIN, false>), dim3(scn_ceil_div(n_samples, kSamplesPerBlock)), dim3(kThreads), lds,
                       st, pts, viewdirs, vd_stride, samples_per_ray, wpacked, raw, save, (long)n_samples, CoarseStage{});
    return scn_launch_status();
}

template <bool TRAIN>
static int launch_coarse_stage(const CoarseStage& cs, const float* wpacked, float* raw, float* save, hipStream_t st) {
    const size_t lds = (size_t)(kStreamBufs * kMaxChunkFwd + Var<3>::kES * kThreads) * sizeof(float);
    const long P = (long)cs.n_rays * kCoarseSamples;
    SCN_LDS_OPT_IN((mlp_fwd_kernel<3, TRAIN, true>), lds);
    hipLaunchKernelGGL((mlp_fwd_kernel<3, TRAIN, true>), dim3(scn_ceil_div(P, kSamplesPerBlock)), dim3(kThreads), lds, st,
                       (const float*)nullptr, cs.rays + 8, cs.ray_stride, kCoarseSamples, wpacked, raw, save, P, cs);
    return scn_launch_status();
}

extern "C" int scnerf_coarse_stage_fwd(const float* rays, int ray_stride, const float* t_vals, const float* t_rand,
                                       int lindisp, const float* wpacked, float* save, const float* noise,
                                       int white_bkgd, float* z, float* pts, float* raw, float* rgb_map,
                                       float* disp_map, float* acc_map, float* depth_map, float* weights, int n_rays,
                                       int n_samples, void* stream) {
    SCN_RETURN_IF(!rays || !t_vals || !wpacked || !z || !pts || !raw || !rgb_map || !disp_map || !acc_map, SCN_EINVAL);
    SCN_RETURN_IF(n_rays < 0 || ray_stride < 11, SCN_EINVAL);
    SCN_RETURN_IF(n_samples != kCoarseSamples, SCN_ENOSUP);           // two wave tiles per ray is what the fusion rests on
    if (n_rays == 0) return 0;
    const CoarseStage cs{rays, ray_stride, n_rays, t_vals, t_rand, lindisp, z, pts, noise, white_bkgd,
                         rgb_map, disp_map, acc_map, depth_map, weights};
    hipStream_t st = (hipStream_t)stream;
    return save ? launch_coarse_stage<true>(cs, wpacked, raw, save, st) : launch_coarse_stage<false>(cs, wpacked, raw, save, st);
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
