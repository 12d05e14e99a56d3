// mlp_bwd.hip -- fused data-gradient chain of the NeRF MLP for gfx950 (the "dgrad" half of the
// backward pass; the weight gradients are GEMMs over the tensors this kernel and the training
// forward leave in HBM: wgrad.hip).
//
// Mirror image of mlp_fwd.hip: a wave owns the same 32 samples as in the forward launch, the
// gradient w.r.t. a layer's output lives in registers in the MFMA accumulator layout and is fed
// straight back as the B operand of the transposed layer; W^T streams through LDS from the
// backward packed buffer (mlp_layout.backward_index()).  ReLU masks come from the lane-native
// bit masks the forward saved (16 bytes per lane and layer).
//
// Gradient of: NeRF.forward + Embedder + run_network
//   /root/reference NeRF/run_nerf_helpers.py:105-128, :24-72, NeRF/create_nerf.py:18-32
// (what autograd derives there).  Outputs: dZ of every layer (row-major, for wgrad), d pts,
// d viewdirs (per sample; summed per ray by ray_reduce).
#include <scn_wave.h>

#include "launch.h"
#include "mlp_common.h"
#include "scnerf_hip.h"

namespace {

using namespace scn;
using namespace scn::mlp;

template <int PD>
__device__ __forceinline__ u32x4 load_mask(const float* save, long P, int section, long wave_tile, int lane) {
    return *reinterpret_cast<const u32x4*>(mask_ptr<PD>(const_cast<float*>(save), P, section, wave_tile, lane));
}

// dst[16 t + r] = bit ? acc[t][r] : 0
template <int NT>
__device__ __forceinline__ void mask_to_regs(const f32x16 (&acc)[NT], u32x4 bits, float (&dst)[16 * NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 16 * t + r;
            dst[i] = mask_select(acc[t][r], bits, i);
        }
}

// Gradient of the positional encoding held in slot layout (mlp_common.h pe_slots): returns this lane
// half's contributions.  PD == 3: (d/d(h ? y : x), d/dz -- both halves carry a z part, the caller adds
// them); PD == 4: (d/d(h ? y : x), d/d(h ? w : z)), complete per half.
template <int PD, int L, int NS>
__device__ __forceinline__ void pe_backward(float x, float y, float z, float w, int h, const float (&de)[NS],
                                            float* d_a, float* d_b) {
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

template <int PD>
__global__ __launch_bounds__(kThreads, 1) void mlp_bwd_kernel(
    const float* __restrict__ d_raw, const float* __restrict__ pts, const float* __restrict__ viewdirs,
    int vd_stride, int samples_per_ray, const float* __restrict__ wbk, const float* __restrict__ save,
    float* __restrict__ grads, float* __restrict__ d_pts, float* __restrict__ d_views, long P) {
    const int lane = lane_id();
    const int m = lane & 31, h = lane >> 5;
    const long wave_tile = (long)blockIdx.x * 4 + wave_id();
    const long p = wave_tile * kSamplesPerWave + m;
    const bool live = p < P;
    const long pc = live ? p : P - 1;
    const long Ppad = padded_samples(P);
    using V = Var<PD>;
    constexpr int ET = V::kET, ECS = V::kECS, ES = V::kES;

    WStream ws;
    // stream order: rgb^T, views^T (8 tiles), its encoded-direction tile, feature^T, 7, 6, 5 main, 5 skip, 4 .. 1, 0
    float dz[128];
    float de[ES];
#pragma unroll
    for (int s = 0; s < ES; ++s) de[s] = 0.f;
    ws.g = reinterpret_cast<const f32x4*>(wbk);
    stream_prime<1>(ws);           // RGBT: 4 tiles x 4 steps = 1024 floats

    // dead lanes (p >= P) must contribute exact zeros: their d_raw is forced to 0
    f32x4 dr = *reinterpret_cast<const f32x4*>(d_raw + pc * 4);
    if (!live) dr = f32x4{0.f, 0.f, 0.f, 0.f};
    const float dsigma = dr[3];

    // ---- rgb_linear^T : d hv = W_rgb^T d rgb  (contraction over the 3 channels) ----------
    float brgb[4] = {h ? dr[1] : dr[0], h ? 0.f : dr[2], 0.f, 0.f};
    f32x16 acc4[4];
    zero_acc<4>(acc4);
    mfma_part<4, 4, 4, 8>(brgb, acc4, ws);
    float dzv[64];
    mask_to_regs<4>(acc4, load_mask<PD>(save, P, 8, wave_tile, lane), dzv);
    // every gradient tensor the wgrad GEMMs need is the B operand of the next part: it is stored
    // chunk by chunk while that part runs (mfma_part's save_row)

    // ---- views layer^T : [d feature | d encoded dir] = W_v^T dZ_v ------------------------
    f32x16 acc[8];
    zero_acc<8>(acc);
    mfma_part<64, 8, 16, 4>(dzv, acc, ws, tile_ptr(grads + (long)kGradDzv * Ppad, wave_tile, 128, lane));
    f32x16 acce1[1];
    zero_acc<1>(acce1);
    mfma_part<64, 1, 64, 8>(dzv, acce1, ws);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[16 * t + r] = acc[t][r];          // d feature (linear layer)
    {
        float dev[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dev[r] = acce1[0][r];
        const long ray = pc / samples_per_ray;
        float gxy, gz;
        pe_backward<3, 4, 16>(viewdirs[ray * vd_stride + 0], viewdirs[ray * vd_stride + 1], viewdirs[ray * vd_stride + 2], 0.f, h, dev, &gxy, &gz);
        const float oxy = shfl_xor(gxy, 32), oz = shfl_xor(gz, 32);
        if (live && h == 0) {
            d_views[p * 3 + 0] = gxy;
            d_views[p * 3 + 1] = oxy;
            d_views[p * 3 + 2] = gz + oz;
        }
    }

    // ---- feature_linear^T + alpha_linear^T : d h8 = W_f^T d feature + w_alpha d sigma -----
    {
        const float* wa = wbk + V::kBwdAlphaW;          // lane-vector layout
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = lane_vec(wa, t, q, h);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][4 * q + j] = w[j] * dsigma;
            }
    }
    mfma_part<128, 8, 16, 8>(dz, acc, ws, tile_ptr(grads + (long)kGradDfeat * Ppad, wave_tile, 256, lane));
    mask_to_regs<8>(acc, load_mask<PD>(save, P, 7, wave_tile, lane), dz);      // dZ of trunk layer 7

    // ---- trunk layers 7..1 : d h_{l-1} = W_l^T dZ_l, then the ReLU mask of layer l-1 ------
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
        zero_acc<8>(acc);
        mfma_part<128, 8, 16, 8>(dz, acc, ws, tile_ptr(grads + (long)(kGradDz + l * 256) * Ppad, wave_tile, 256, lane));
        if (l == 5) {
            // skip connection: layer 5 also consumed the encoded point (its first columns).  The PD == 4
            // stream carries a 4th, all-zero tile so that every chunk stays 8192 floats.
            f32x16 acce[ET];
            zero_acc<ET>(acce);
            mfma_part<128, ET, ECS, 8>(dz, acce, ws);
#pragma unroll
            for (int t = 0; t < ES / 16; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) de[16 * t + r] = acce[t][r];
        }
        mask_to_regs<8>(acc, load_mask<PD>(save, P, l - 1, wave_tile, lane), dz);
    }

    // ---- layer 0^T : d encoded point, then the encoding's own gradient -> d pts ------------
    {
        f32x16 acce[ET];
        zero_acc<ET>(acce);
        mfma_part<128, ET, ECS, 0>(dz, acce, ws, tile_ptr(grads + (long)kGradDz * Ppad, wave_tile, 256, lane));
#pragma unroll
        for (int t = 0; t < ES / 16; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) de[16 * t + r] += acce[t][r];
        float ga, gb;
        pe_backward<PD, 10, ES>(pts[pc * PD + 0], pts[pc * PD + 1], pts[pc * PD + 2],
                                PD == 4 ? pts[pc * PD + (PD - 1)] : 0.f, h, de, &ga, &gb);
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

}  // namespace

template <int PD>
static int launch_bwd(const float* d_raw, const float* pts, const float* viewdirs, int vd_stride, int samples_per_ray,
                      const float* wpacked_bwd, const float* save, float* grads, float* d_pts, float* d_views,
                      long long n_samples, hipStream_t st) {
    const size_t lds = (size_t)kStreamBufs * kMaxChunkBwd * sizeof(float);      // 96 KB: needs the opt-in
    SCN_LDS_OPT_IN((mlp_bwd_kernel<PD>), lds);
    hipLaunchKernelGGL((mlp_bwd_kernel<PD>), dim3(scn_ceil_div(n_samples, kSamplesPerBlock)), dim3(kThreads), lds, st,
                       d_raw, pts, viewdirs, vd_stride, samples_per_ray, wpacked_bwd, save, grads, d_pts, d_views,
                       (long)n_samples);
    return scn_launch_status();
}

extern "C" int scnerf_mlp_bwd(int pt_dims, const float* d_raw, const float* pts, const float* viewdirs,
                              int vd_stride, int samples_per_ray, const float* wpacked_bwd, const float* save,
                              float* grads, float* d_pts, float* d_views, long long n_samples,
                              void* stream) {
    SCN_RETURN_IF(!d_raw || !pts || !viewdirs || !wpacked_bwd || !save || !grads || !d_pts || !d_views, SCN_EINVAL);
    SCN_RETURN_IF(samples_per_ray < 1 || vd_stride < 3 || n_samples < 0 || (pt_dims != 3 && pt_dims != 4), SCN_EINVAL);
    if (n_samples == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    return pt_dims == 3 ? launch_bwd<3>(d_raw, pts, viewdirs, vd_stride, samples_per_ray, wpacked_bwd, save, grads, d_pts, d_views, n_samples, st)
                        : launch_bwd<4>(d_raw, pts, viewdirs, vd_stride, samples_per_ray, wpacked_bwd, save, grads, d_pts, d_views, n_samples, st);
}
