// composite.hip -- alpha compositing of the network output along each ray, forward + backward,
// and the per-ray reduction of the point gradients.  One 64-lane wave owns one ray and walks it in
// passes of 64 samples; the transmittance is a wave prefix product carried across passes.
//
// Replaces raw2outputs (/root/reference NeRF/render.py:302-355) and what autograd derives from it.
// Numerics follow the reference's op-by-op fp32 arithmetic (file built with -ffp-contract=off);
// the running product is kept in fp64 and rounded per sample, which is what ATen's CPU cumprod
// does for float, and the three weighted sums are accumulated in fp64.
#include <scn_wave.h>

#include "launch.h"
#include "ray_stage.h"
#include "scnerf_hip.h"

namespace {

using namespace scn;

constexpr int kRaysPerBlock = 4;

__device__ __forceinline__ double wave_incl_sum_rev(double v, int lane) {
    // inclusive suffix sum: v[lane] + v[lane+1] + ... + v[63]
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = shfl_down(v, o);
        if (lane + o < 64) v += u;
    }
    return v;
}

using ray::ray_norm;
using ray::SampleTerms;
using ray::sample_terms;
using ray::sigmoidf;
using ray::wave_incl_prod;
using ray::wave_sum;

__global__ __launch_bounds__(256) void composite_fwd_kernel(
    const float* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rays, int ray_stride,
    const float* __restrict__ noise, int white_bkgd, float* __restrict__ rgb_map, float* __restrict__ disp_map,
    float* __restrict__ acc_map, float* __restrict__ depth_map, float* __restrict__ weights, int n, int s) {
    const int lane = lane_id();
    int ray = blockIdx.x * kRaysPerBlock + wave_id();
    const bool live = ray < n;
    if (!live) ray = n - 1;
    const float norm = ray_norm(rays + (size_t)ray * ray_stride + 3);
    const float* rr = raw + (size_t)ray * s * 4;
    const float* zr = z + (size_t)ray * s;
    auto fetch = [&](int i, f32x4* rw, float* zi) {
        *rw = *reinterpret_cast<const f32x4*>(rr + (size_t)i * 4);
        *zi = zr[i];
    };
    ray::composite_ray(fetch, s, norm, noise ? noise + (size_t)ray * s : nullptr, white_bkgd, live, lane,
                       rgb_map + (size_t)ray * 3, disp_map + ray, acc_map + ray, depth_map ? depth_map + ray : nullptr,
                       weights ? weights + (size_t)ray * s : nullptr);
}

// Backward of the above.  g_* are the incoming gradients of the four maps (any may be NULL);
// g_raw_in is an optional gradient arriving directly at `raw` (the retraw output).
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rays, int ray_stride,
    const float* __restrict__ noise, int white_bkgd, const float* __restrict__ g_rgb,
    const float* __restrict__ g_disp, const float* __restrict__ g_acc, const float* __restrict__ g_depth,
    const float* __restrict__ g_raw_in, float* __restrict__ d_raw, float* __restrict__ d_rays_d, int n, int s,
    int lds_per_wave) {
    float* lds = dynamic_lds<float>() + (size_t)wave_id() * lds_per_wave;
    float* s_T = lds;          // transmittance per sample
    const int lane = lane_id();
    int ray = blockIdx.x * kRaysPerBlock + wave_id();
    const bool live = ray < n;
    if (!live) ray = n - 1;
    const float* rd = rays + (size_t)ray * ray_stride + 3;
    const float norm = ray_norm(rd);
    const float* zr = z + (size_t)ray * s;

    // sweep 1 (front to back): transmittance per sample, depth and opacity totals
    double carry = 1.0, sdepth = 0.0, sacc = 0.0;
    for (int base = 0; base < s; base += 64) {
        const int i = base + lane;
        const bool in = i < s;
        const int ic = in ? i : s - 1;
        const float sigma = raw[((size_t)ray * s + ic) * 4 + 3];
        const float zi = zr[ic];
        const float zn = ic + 1 < s ? zr[ic + 1] : zi;
        const float nz = noise ? noise[(size_t)ray * s + ic] : 0.f;
        const SampleTerms t = sample_terms(sigma, nz, zi, zn, ic == s - 1, norm);
        const double incl = wave_incl_prod(in ? (double)t.q : 1.0, lane) * carry;
        double excl = shfl_up(incl, 1);
        if (lane == 0) excl = carry;
        carry = shfl(incl, 63);
        const float T = (float)excl;
        if (in) {
            s_T[i] = T;
            const float w = t.alpha * T;
            sdepth += (double)(w * zi);
            sacc += (double)w;
        }
    }
    sdepth = wave_sum(sdepth);
    sacc = wave_sum(sacc);
    block_sync();

    const float acc = (float)sacc, depth = (float)sdepth;
    float gr = 0.f, gg = 0.f, gb = 0.f;
    if (g_rgb) { gr = g_rgb[(size_t)ray * 3]; gg = g_rgb[(size_t)ray * 3 + 1]; gb = g_rgb[(size_t)ray * 3 + 2]; }
    float gdepth = g_depth ? g_depth[ray] : 0.f;
    float gacc = g_acc ? g_acc[ray] : 0.f;
    if (g_disp) {
        const float den = acc + 1e-10f;
        const float q = depth / den;
        if (q > 1e-10f) {
            const float gq = -g_disp[ray] / (q * q);     // d(1/q)
            gdepth += gq / den;
            gacc += gq * (-depth / (den * den));
        }
    }
    if (white_bkgd) gacc -= gr + gg + gb;

    // sweep 2 (back to front): suffix sums of G_k w_k, per-sample gradients
    double suffix = 0.0;     // sum over samples behind the current pass
    double dnorm = 0.0;
    const int npass = (s + 63) / 64;
    for (int pass = npass - 1; pass >= 0; --pass) {
        const int i = pass * 64 + lane;
        const bool in = i < s;
        const int ic = in ? i : s - 1;
        const f32x4 rw = *reinterpret_cast<const f32x4*>(raw + ((size_t)ray * s + ic) * 4);
        const float zi = zr[ic];
        const float zn = ic + 1 < s ? zr[ic + 1] : zi;
        const float nz = noise ? noise[(size_t)ray * s + ic] : 0.f;
        const SampleTerms t = sample_terms(rw[3], nz, zi, zn, ic == s - 1, norm);
        const float T = s_T[ic];
        const float w = t.alpha * T;
        const float c0 = sigmoidf(rw[0]), c1 = sigmoidf(rw[1]), c2 = sigmoidf(rw[2]);
        const float G = gr * c0 + gg * c1 + gb * c2 + gdepth * zi + gacc;     // dL/dw_i
        const double gw = in ? (double)G * (double)w : 0.0;
        const double incl = wave_incl_sum_rev(gw, lane) + suffix;   // sum_{k >= i}
        const double after = incl - gw;                               // sum_{k > i}
        suffix = shfl(incl, 0);
        if (in) {
            const float dalpha = (float)((double)G * (double)T - after / (double)t.q);
            const bool on = (rw[3] + nz) > 0.f;
            f32x4 o;
            o[0] = gr * w * c0 * (1.f - c0);
            o[1] = gg * w * c1 * (1.f - c1);
            o[2] = gb * w * c2 * (1.f - c2);
            o[3] = on ? dalpha * t.dist * t.e : 0.f;
            if (g_raw_in) {
                const f32x4 gi = *reinterpret_cast<const f32x4*>(g_raw_in + ((size_t)ray * s + i) * 4);
                o[0] += gi[0]; o[1] += gi[1]; o[2] += gi[2]; o[3] += gi[3];
            }
            if (live) *reinterpret_cast<f32x4*>(d_raw + ((size_t)ray * s + i) * 4) = o;
            dnorm += (double)(dalpha * t.a * t.e * t.draw);           // via dist = draw * |d|
        }
    }
    dnorm = wave_sum(dnorm);
    if (live && lane == 0 && d_rays_d) {
        const float k = norm > 0.f ? (float)dnorm / norm : 0.f;
        d_rays_d[(size_t)ray * 3 + 0] = k * rd[0];
        d_rays_d[(size_t)ray * 3 + 1] = k * rd[1];
        d_rays_d[(size_t)ray * 3 + 2] = k * rd[2];
    }
}

// d ray_batch[:, 0:3] (+)= sum_s d_pts;  [:, 3:6] (+)= sum_s d_pts * z (+ extra_d);
// [:, 8:11] (+)= sum_s d_views.  pts = o + d z (NeRF/render.py:259, :277), viewdirs broadcast
// over the samples (create_nerf.py:25).  Columns 6:8 (near, far) get no gradient.
__global__ __launch_bounds__(256) void ray_reduce_kernel(
    const float* __restrict__ d_pts, const float* __restrict__ d_views, const float* __restrict__ z,
    const float* __restrict__ extra_d, float* __restrict__ d_rays, int ray_stride, int accumulate, int n,
    int s) {
    const int lane = lane_id();
    int ray = blockIdx.x * kRaysPerBlock + wave_id();
    if (ray >= n) return;
    double o[3] = {0, 0, 0}, d[3] = {0, 0, 0}, v[3] = {0, 0, 0};
    for (int i = lane; i < s; i += 64) {
        const size_t k = (size_t)ray * s + i;
        const float zi = z[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = d_pts[k * 3 + c];
            o[c] += (double)g;
            d[c] += (double)(g * zi);
            if (d_views) v[c] += (double)d_views[k * 3 + c];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { o[c] = wave_sum(o[c]); d[c] = wave_sum(d[c]); v[c] = wave_sum(v[c]); }
    if (lane == 0) {
        float* r = d_rays + (size_t)ray * ray_stride;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ex = extra_d ? extra_d[(size_t)ray * 3 + c] : 0.f;
            const float go = (float)o[c], gd = (float)d[c] + ex, gv = (float)v[c];
            if (accumulate) {
                r[c] += go; r[3 + c] += gd;
                if (ray_stride > 8) r[8 + c] += gv;
            } else {
                r[c] = go; r[3 + c] = gd;
                if (ray_stride > 8) r[8 + c] = gv;
            }
        }
        if (!accumulate) { r[6] = 0.f; r[7] = 0.f; }
    }
}

}  // namespace

extern "C" int scnerf_composite_fwd(const float* raw, const float* z, const float* rays, int ray_stride,
                                    const float* noise, int white_bkgd, float* rgb_map, float* disp_map,
                                    float* acc_map, float* depth_map, float* weights, int n, int s,
                                    void* stream) {
    SCN_RETURN_IF(!raw || !z || !rays || !rgb_map || !disp_map || !acc_map || n < 0 || s < 1 || ray_stride < 6, SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(scn_ceil_div(n, kRaysPerBlock)), dim3(256), 0,
                       (hipStream_t)stream, raw, z, rays, ray_stride, noise, white_bkgd, rgb_map, disp_map,
                       acc_map, depth_map, weights, n, s);
    return scn_launch_status();
}

extern "C" int scnerf_composite_bwd(const float* raw, const float* z, const float* rays, int ray_stride,
                                    const float* noise, int white_bkgd, const float* g_rgb,
                                    const float* g_disp, const float* g_acc, const float* g_depth,
                                    const float* g_raw_in, float* d_raw, float* d_rays_d, int n, int s,
                                    void* stream) {
    SCN_RETURN_IF(!raw || !z || !rays || !d_raw || n < 0 || s < 1 || ray_stride < 6, SCN_EINVAL);
    if (n == 0) return 0;
    const int per_wave = (s + 3) / 4 * 4;
    const size_t lds = (size_t)per_wave * 4 * kRaysPerBlock;
    SCN_RETURN_IF(lds > 64 * 1024, SCN_ENOSUP);
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(scn_ceil_div(n, kRaysPerBlock)), dim3(256), lds,
                       (hipStream_t)stream, raw, z, rays, ray_stride, noise, white_bkgd, g_rgb, g_disp, g_acc,
                       g_depth, g_raw_in, d_raw, d_rays_d, n, s, per_wave);
    return scn_launch_status();
}

extern "C" int scnerf_ray_reduce(const float* d_pts, const float* d_views, const float* z,
                                 const float* extra_d, float* d_rays, int ray_stride, int accumulate, int n,
                                 int s, void* stream) {
    SCN_RETURN_IF(!d_pts || !z || !d_rays || n < 0 || s < 1 || ray_stride < 8, SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(ray_reduce_kernel, dim3(scn_ceil_div(n, kRaysPerBlock)), dim3(256), 0,
                       (hipStream_t)stream, d_pts, d_views, z, extra_d, d_rays, ray_stride, accumulate, n, s);
    return scn_launch_status();
}
