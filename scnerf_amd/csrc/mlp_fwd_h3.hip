// mlp_fwd_h3.hip -- the NeRF MLP forward in the "resident" arithmetic (mlp_h3.h): positional encoding -> layer 0 ->
// trunk with skip -> density head, feature layer, views layer, rgb -- and for the coarse stage the stratified sampling
// in front and the alpha compositing behind -- as ONE launch on three fp16 products per product, the activations
// register-resident as cut fp16 planes from the encoding to the colour.  Training mode writes every activation the
// weight-gradient GEMMs and the data-gradient chain need (the workspace of mlp_common.h, same layout as mlp_fwd.hip's)
// and reads none of them back.
//
// Replaces, for the standard network: run_network + Embedder + NeRF.forward
//   /root/reference NeRF/create_nerf.py:18-32, NeRF/run_nerf_helpers.py:24-72, :105-128
// and for the coarse stage NeRF/render.py:235-262 (sampling, raw2outputs).
#include <scn_wave.h>

#include "launch.h"
#include "mlp_fwd_h3_api.h"
#include "mlp_h3.h"
#include "scnerf_hip.h"

namespace {

using namespace scn;
using namespace scn::h3;

// ---- packing: flat parameters -> fp16 planes in fragment order, and the per-layer scales --------------------------------
// jobs [12][4]: (weight offset, rows, columns, bias offset); table [12][8]: Sw, 1 / Sw, A, B, A'.  Two launches, every sum
// in a fixed order (the scales must not change from run to run):
//   h3_scale_partial_kernel  grid (12 layers, 16 row blocks): a block's 16 rows (a wave per row, coalesced along it) ->
//                            the block's largest |w|, largest row 1-norm, largest |b| and its partial column sums
//   h3_scale_finish_kernel   grid (12): column sums = the 16 partials added in order; the layer's five numbers
// A parameter that is not finite (a diverged run) must stay visible at the loss as it does in the reference, where
// torch.relu and the matrix products carry a NaN through to the colours: the kernels' v_max-based ReLU and maxima drop
// NaNs, so the scale pass looks at every weight and bias anyway and, if one is NaN or infinite, leaves a NaN in the rgb
// layer's kPoison slot, which the forward kernel adds to the rgb layer's output scale -- every colour of the launch is NaN.
constexpr int kScaleBlocks = 16, kScaleCols = 512;
constexpr int kScalePartial = 4 + kScaleCols;      // floats per (layer, row block)

__global__ __launch_bounds__(256) void h3_scale_partial_kernel(const float* __restrict__ params, const int* __restrict__ jobs,
                                                               float* __restrict__ partial, float* __restrict__ table) {
    float* red = dynamic_lds<float>();              // [4 waves][4] + [4 waves][512] column partials
    float* cols_w = red + 16;
    const int l = blockIdx.x, blk = blockIdx.y, tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const float* wt = params + jobs[4 * l];
    const int rows = jobs[4 * l + 1], cols = jobs[4 * l + 2];
    const float* bs = params + jobs[4 * l + 3];
    const int per = (rows + kScaleBlocks - 1) / kScaleBlocks;
    const int r0 = blk * per, r1 = min(rows, r0 + per);
    if (l == kLayerRgb && blk == 0 && tid == 0) table[kLayerRgb * kScaleStride + kPoison] = 0.f;     // (the finish pass may set it)
    float mx = 0.f, rowsum = 0.f, bmax = 0.f, bad = 0.f;
    float colp[kScaleCols / 64];
#pragma unroll
    for (int k = 0; k < kScaleCols / 64; ++k) colp[k] = 0.f;
    for (int r = r0 + wave; r < r1; r += 4) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kScaleCols / 64; ++k) {
            const int c = lane + 64 * k;
            const float a = c < cols ? fabsf(wt[(long)r * cols + c]) : 0.f;
            s += a;
            mx = fmaxf(mx, a);
            colp[k] += a;
        }
        for (int o = 32; o > 0; o >>= 1) s += shfl_xor(s, o);
        rowsum = fmaxf(rowsum, s);
        bmax = fmaxf(bmax, fabsf(bs[r]));
        // (a NaN or an infinity among the row's weights is in its sum; fmaxf drops NaNs)
        if (!(s < __builtin_huge_valf()) || !(fabsf(bs[r]) < __builtin_huge_valf())) bad = 1.f;
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, shfl_xor(mx, o));
#pragma unroll
    for (int k = 0; k < kScaleCols / 64; ++k) cols_w[wave * kScaleCols + lane + 64 * k] = colp[k];
    if (lane == 0) { red[wave * 4] = mx; red[wave * 4 + 1] = rowsum; red[wave * 4 + 2] = bmax; red[wave * 4 + 3] = bad; }
    block_sync();
    float* out = partial + ((long)l * kScaleBlocks + blk) * kScalePartial;
    if (tid < 4) out[tid] = fmaxf(fmaxf(red[tid], red[4 + tid]), fmaxf(red[8 + tid], red[12 + tid]));
    for (int c = tid; c < kScaleCols; c += 256)
        out[4 + c] = ((cols_w[c] + cols_w[kScaleCols + c]) + cols_w[2 * kScaleCols + c]) + cols_w[3 * kScaleCols + c];
}

__global__ __launch_bounds__(256) void h3_scale_finish_kernel(const float* __restrict__ partial, const int* __restrict__ jobs,
                                                              float* __restrict__ table) {
    float* red = dynamic_lds<float>();              // 256 floats
    const int l = blockIdx.x, tid = threadIdx.x;
    const int cols = jobs[4 * l + 2];
    const float* p = partial + (long)l * kScaleBlocks * kScalePartial;
    float colsum = 0.f;
    for (int c = tid; c < cols; c += 256) {
        float s = 0.f;
        for (int b = 0; b < kScaleBlocks; ++b) s += p[b * kScalePartial + 4 + c];
        colsum = fmaxf(colsum, s);
    }
    red[tid] = colsum;
    block_sync();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        block_sync();
    }
    if (tid == 0) {
        float mx = 0.f, rowsum = 0.f, bmax = 0.f, bad = 0.f;
        for (int b = 0; b < kScaleBlocks; ++b) {
            mx = fmaxf(mx, p[b * kScalePartial]);
            rowsum = fmaxf(rowsum, p[b * kScalePartial + 1]);
            bmax = fmaxf(bmax, p[b * kScalePartial + 2]);
            bad = fmaxf(bad, p[b * kScalePartial + 3]);
        }
        if (bad > 0.f) table[kLayerRgb * kScaleStride + kPoison] = __builtin_nanf("");
        const float sw = scn::h3::scale_for(mx);
        float* t = table + l * kScaleStride;
        t[kSw] = sw;
        t[kSwInv] = scn::h3::inv_pow2(sw);
        // (sums of up to 339 fp32 terms: a relative 1e-4 covers their rounding)
        t[kBoundA] = rowsum * 1.0001f;
        t[kBoundB] = bmax;
        t[kBoundAT] = red[0] * 1.0001f;
        t[5] = mx; t[6] = 0.f;
        if (l != kLayerRgb) t[kPoison] = 0.f;
    }
}

// one thread per fragment element: idx [n_frags * 512] (flat parameter index or -1), meta [n_frags] (plane | layer << 1)
__global__ __launch_bounds__(256) void h3_pack_kernel(const float* __restrict__ params, const int* __restrict__ idx,
                                                      const unsigned char* __restrict__ meta, const float* __restrict__ table,
                                                      short* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = idx[i];
    const unsigned mt = meta[i >> 9];
    const float x = j >= 0 ? params[j] * table[(mt >> 1) * kScaleStride + kSw] : 0.f;
    const unsigned hb = f16_bits(x);
    out[i] = (short)((mt & 1u) ? f16_bits(x - f16_value(hb)) : hb);
}

}  // namespace

// the scale table, followed by the scale pass's own scratch (per layer and row block: three maxima + column partials)
extern "C" long long scnerf_h3_scale_floats(void) {
    return (long long)kScaleLayers * kScaleStride + (long long)kScaleLayers * kScaleBlocks * kScalePartial;
}

extern "C" int scnerf_h3_pack(const float* flat_params, const int* jobs, const int* idx_fwd, const unsigned char* meta_fwd,
                              long long frags_fwd, const int* idx_bwd, const unsigned char* meta_bwd, long long frags_bwd,
                              short* stream_fwd, short* stream_bwd, float* scales, void* stream) {
    SCN_RETURN_IF(!flat_params || !jobs || !scales, SCN_EINVAL);
    SCN_RETURN_IF((frags_fwd > 0 && (!idx_fwd || !meta_fwd || !stream_fwd)) || (frags_bwd > 0 && (!idx_bwd || !meta_bwd || !stream_bwd)), SCN_EINVAL);
    SCN_RETURN_IF(frags_fwd < 0 || frags_bwd < 0, SCN_EINVAL);
    hipStream_t st = (hipStream_t)stream;
    float* partial = scales + kScaleLayers * kScaleStride;
    hipLaunchKernelGGL(h3_scale_partial_kernel, dim3(kScaleLayers, kScaleBlocks), dim3(256), (16 + 4 * kScaleCols) * 4, st,
                       flat_params, jobs, partial, scales);
    hipLaunchKernelGGL(h3_scale_finish_kernel, dim3(kScaleLayers), dim3(256), 1024, st, partial, jobs, scales);
    if (frags_fwd > 0)
        hipLaunchKernelGGL(h3_pack_kernel, dim3(scn_ceil_div(frags_fwd * 512, 256)), dim3(256), 0, st, flat_params, idx_fwd,
                           meta_fwd, scales, stream_fwd, (long)(frags_fwd * 512));
    if (frags_bwd > 0)
        hipLaunchKernelGGL(h3_pack_kernel, dim3(scn_ceil_div(frags_bwd * 512, 256)), dim3(256), 0, st, flat_params, idx_bwd,
                           meta_bwd, scales, stream_bwd, (long)(frags_bwd * 512));
    return scn_launch_status();
}

extern "C" int scnerf_mlp_fwd_h3(int pt_dims, const float* pts, const float* viewdirs, int vd_stride, int samples_per_ray,
                                 const float* wpacked, const short* stream_fwd, const float* scales, float* raw, float* save,
                                 long long n_samples, float* chunk_amax, int n_chunks, long long chunk_samples, void* stream) {
    SCN_RETURN_IF(!pts || !viewdirs || !wpacked || !stream_fwd || !scales || !raw, SCN_EINVAL);
    SCN_RETURN_IF(samples_per_ray < 1 || vd_stride < 3 || n_samples < 0 || (pt_dims != 3 && pt_dims != 4), SCN_EINVAL);
    SCN_RETURN_IF(n_samples >= (1LL << 31), SCN_ENOSUP);       // (the kernels index samples with 31 bits)
    SCN_RETURN_IF(chunk_amax && (n_chunks < 1 || chunk_samples < 32 || chunk_samples % 32), SCN_EINVAL);
    if (n_samples == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const scn::h3f::ChunkMaxima cm{chunk_amax, n_chunks, (long)chunk_samples};
    return pt_dims == 3 ? scn::h3f::fwd_h3_pd3(pts, viewdirs, vd_stride, samples_per_ray, wpacked, stream_fwd, scales, raw, save, n_samples, cm, st)
                        : scn::h3f::fwd_h3_pd4(pts, viewdirs, vd_stride, samples_per_ray, wpacked, stream_fwd, scales, raw, save, n_samples, cm, st);
}

extern "C" int scnerf_coarse_stage_fwd_h3(const float* rays, int ray_stride, const float* t_vals, const float* t_rand,
                                          int lindisp, const float* wpacked, const short* stream_fwd, const float* scales,
                                          float* save, const float* noise, int white_bkgd, float* z, float* pts, float* raw,
                                          float* rgb_map, float* disp_map, float* acc_map, float* depth_map, float* weights,
                                          int n_rays, int n_samples, float* chunk_amax, int n_chunks, long long chunk_samples,
                                          void* stream) {
    SCN_RETURN_IF(!rays || !t_vals || !wpacked || !stream_fwd || !scales || !z || !pts || !raw || !rgb_map || !disp_map || !acc_map, SCN_EINVAL);
    SCN_RETURN_IF(n_rays < 0 || ray_stride < 11, SCN_EINVAL);
    SCN_RETURN_IF(n_samples != scn::h3f::kCoarseSamples || n_rays >= (1 << 25), SCN_ENOSUP);     // (31-bit sample indices)
    SCN_RETURN_IF(chunk_amax && (n_chunks < 1 || chunk_samples < 32 || chunk_samples % 32), SCN_EINVAL);
    if (n_rays == 0) return 0;
    const scn::h3f::CoarseStage cs{rays, ray_stride, n_rays, t_vals, t_rand, lindisp, z, pts, noise, white_bkgd,
                                   rgb_map, disp_map, acc_map, depth_map, weights};
    const scn::h3f::ChunkMaxima cm{chunk_amax, n_chunks, (long)chunk_samples};
    return scn::h3f::fwd_h3_coarse(cs, rays, ray_stride, wpacked, stream_fwd, scales, raw, save, cm, (hipStream_t)stream);
}

extern "C" int scnerf_fine_stage_fwd_h3(const float* rays, int ray_stride, const float* z_c, const float* w_c, const float* u,
                                        int u_row_stride, const float* wpacked, const short* stream_fwd, const float* scales,
                                        float* save, const float* noise, int white_bkgd, float* z_f, float* pts_f,
                                        float* z_samples, float* z_std, long long* inds, float* cdf, float* raw, float* rgb_map,
                                        float* disp_map, float* acc_map, float* depth_map, float* weights, int n_rays,
                                        int n_coarse, int n_importance, float* chunk_amax, int n_chunks,
                                        long long chunk_samples, void* stream) {
    SCN_RETURN_IF(!rays || !z_c || !w_c || !u || !wpacked || !stream_fwd || !scales, SCN_EINVAL);
    SCN_RETURN_IF(!z_f || !pts_f || !z_samples || !z_std || !raw || !rgb_map || !disp_map || !acc_map, SCN_EINVAL);
    SCN_RETURN_IF(n_rays < 0 || ray_stride < 11 || (u_row_stride != 0 && u_row_stride != n_importance), SCN_EINVAL);
    // a workgroup takes whole rays through passes of 128 samples: 64 + N_importance in {128, 192, 256}
    SCN_RETURN_IF(n_coarse != scn::h3f::kCoarseSamples || (n_importance != 64 && n_importance != 128 && n_importance != 192), SCN_ENOSUP);
    SCN_RETURN_IF((long long)n_rays * (n_coarse + n_importance) >= (1LL << 31), SCN_ENOSUP);     // (31-bit sample indices)
    SCN_RETURN_IF(chunk_amax && (n_chunks < 1 || chunk_samples < 32 || chunk_samples % 32), SCN_EINVAL);
    if (n_rays == 0) return 0;
    const int tot = n_coarse + n_importance;
    const int rays_per_block = tot == 192 ? 2 : 1, tiles = tot * rays_per_block / 128;
    const scn::h3f::FineStage fs{rays, ray_stride, n_rays, z_c, w_c, u, u_row_stride, n_importance, z_f, pts_f, z_samples, z_std,
                                 reinterpret_cast<int64_t*>(inds), cdf, noise, white_bkgd, rgb_map, disp_map, acc_map,
                                 depth_map, weights, rays_per_block, tiles};
    const scn::h3f::ChunkMaxima cm{chunk_amax, n_chunks, (long)chunk_samples};
    return save ? scn::h3f::fwd_h3_fine_train(fs, wpacked, stream_fwd, scales, raw, save, cm, (hipStream_t)stream)
                : scn::h3f::fwd_h3_fine_infer(fs, wpacked, stream_fwd, scales, raw, cm, (hipStream_t)stream);
}
