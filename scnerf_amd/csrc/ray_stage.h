// ray_stage.h -- per-ray arithmetic shared by the stand-alone kernels (sampling.hip: coarse_sample,
// composite.hip: composite_fwd) and the fused coarse stage (mlp_fwd.hip: coarse_stage instantiation), so that
// the two routes are the SAME instructions and give bit-identical results.
//
//   stratified depths     /root/reference NeRF/render.py:235-257
//   alpha compositing     /root/reference NeRF/render.py:302-355  (raw2outputs)
#pragma once
#include <scn_wave.h>

namespace scn {
namespace ray {

// ---- stratified depths --------------------------------------------------------------------------------
__device__ __forceinline__ float coarse_depth(float near, float far, float t, int lindisp) {
    if (!lindisp) return near * (1.f - t) + far * t;
    return 1.f / (1.f / near * (1.f - t) + 1.f / far * t);
}

// depth of sample i of s along a ray with bounds (near, far); jitter = the ray's uniform variate for this
// sample, or nullptr-equivalent `has_jitter = false` for the bin centres (perturb == 0)
__device__ __forceinline__ float coarse_z(float near, float far, const float* __restrict__ t_vals, int i, int s,
                                          int lindisp, bool has_jitter, float jitter) {
    float z = coarse_depth(near, far, t_vals[i], lindisp);
    if (has_jitter) {
        float lower = z, upper = z;
        if (i > 0) lower = 0.5f * (z + coarse_depth(near, far, t_vals[i - 1], lindisp));
        if (i < s - 1) upper = 0.5f * (coarse_depth(near, far, t_vals[i + 1], lindisp) + z);
        z = lower + (upper - lower) * jitter;
    }
    return z;
}

// ---- compositing ----------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_incl_prod(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = shfl_up(v, o);
        if (lane >= o) v *= u;
    }
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float ray_norm(const float* d) {
    return sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}

struct SampleTerms {
    float e;       // exp(-relu(sigma + noise) * dist)
    float alpha;   // 1 - e
    float q;       // 1 - alpha + 1e-10
    float a;       // relu(sigma + noise)
    float dist;    // (z[i+1] - z[i] | 1e10) * |d|
    float draw;    // z[i+1] - z[i] | 1e10
};

__device__ __forceinline__ SampleTerms sample_terms(float sigma, float noise, float z, float z_next,
                                                    bool last, float norm) {
    SampleTerms t;
    t.draw = last ? 1e10f : (z_next - z);
    t.dist = t.draw * norm;
    t.a = fmaxf(sigma + noise, 0.f);
    t.e = expf(-t.a * t.dist);
    t.alpha = 1.f - t.e;
    t.q = 1.f - t.alpha + 1e-10f;
    return t;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// One wave composites one ray of s samples in passes of 64.  FETCH(i) -> (raw rgb-sigma of sample i, z[i]);
// it is asked for i and, for i + 1 < s, i + 1 (the next depth).  Outputs are written by lane 0 (maps) and by
// every lane (weights) when `live`.
template <class FETCH>
__device__ __forceinline__ void composite_ray(FETCH fetch, int s, float norm, const float* __restrict__ noise_row,
                                              int white_bkgd, bool live, int lane, float* __restrict__ rgb3,
                                              float* __restrict__ disp, float* __restrict__ acc_out,
                                              float* __restrict__ depth_out, float* __restrict__ weights_row) {
    double carry = 1.0;
    double sr = 0.0, sg = 0.0, sb = 0.0, sdepth = 0.0, sacc = 0.0;
    for (int base = 0; base < s; base += 64) {
        const int i = base + lane;
        const bool in = i < s;
        const int ic = in ? i : s - 1;
        f32x4 rw;
        float zi;
        fetch(ic, &rw, &zi);
        float zn = zi;
        if (ic + 1 < s) { f32x4 unused; fetch(ic + 1, &unused, &zn); }
        const float nz = noise_row ? noise_row[ic] : 0.f;
        const SampleTerms t = sample_terms(rw[3], nz, zi, zn, ic == s - 1, norm);
        const double qd = in ? (double)t.q : 1.0;
        const double incl = wave_incl_prod(qd, lane) * carry;
        // exclusive product = inclusive of the previous lane (carry for lane 0)
        double excl = shfl_up(incl, 1);
        if (lane == 0) excl = carry;
        carry = shfl(incl, 63);
        const float T = (float)excl;
        const float w = t.alpha * T;
        if (in) {
            if (live && weights_row) weights_row[i] = w;
            sr += (double)(w * sigmoidf(rw[0]));
            sg += (double)(w * sigmoidf(rw[1]));
            sb += (double)(w * sigmoidf(rw[2]));
            sdepth += (double)(w * zi);
            sacc += (double)w;
        }
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb);
    sdepth = wave_sum(sdepth); sacc = wave_sum(sacc);
    if (live && lane == 0) {
        const float acc = (float)sacc, depth = (float)sdepth;
        float r = (float)sr, g = (float)sg, b = (float)sb;
        if (white_bkgd) {
            const float bg = 1.f - acc;
            r += bg; g += bg; b += bg;
        }
        rgb3[0] = r; rgb3[1] = g; rgb3[2] = b;
        const float q = depth / (acc + 1e-10f);
        *disp = 1.f / fmaxf(1e-10f, q);
        *acc_out = acc;
        if (depth_out) *depth_out = depth;
    }
}

}  // namespace ray
}  // namespace scn
