// nerfpp.hip -- the per-ray pieces of the NeRF++ path of SCNeRF around the fused MLP kernels (SURVEY 8a
// row A17): sphere intersection, stratified jitter, NeRF++'s comparison-count inverse-CDF sampler,
// sample placement (foreground points and the inverted-sphere background parameterisation), the
// two-level compositing -- each with the backward pass autograd derives in the reference.
//
//   /root/reference nerfplusplus/ddp_train_nerf.py:50-132  (intersect_sphere, perturb_samples, sample_pdf)
//   /root/reference nerfplusplus/ddp_model.py:16-45, :74-143 (depth2pts_outside, NerfNet.forward)
//
// One 64-lane wave owns one ray.  Arithmetic is rounded op by op (-ffp-contract=off) like the reference's
// fp32 tensor code; the pdf normaliser uses ATen's row-sum order, running products / prefix sums are
// carried in fp64 and rounded per element as ATen's CPU cumprod / cumsum do for float.
#include <scn_wave.h>

#include "aten_sum.h"
#include "launch.h"
#include "scnerf_hip.h"

namespace {

using namespace scn;

constexpr int kRaysPerBlock = 4;
constexpr float kTiny = 1e-6f;     // nerfplusplus/utils.py:8
constexpr float kHuge = 1e10f;     // nerfplusplus/utils.py:7

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ V3 ld3(const float* p, long i) { return mk(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]); }
__device__ __forceinline__ void st3(float* p, long i, V3 v) { p[i * 3] = v.x; p[i * 3 + 1] = v.y; p[i * 3 + 2] = v.z; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_incl_prod(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = shfl_up(v, o);
        if (lane >= o) v *= u;
    }
    return v;
}
__device__ __forceinline__ double wave_incl_sum_rev(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = shfl_down(v, o);
        if (lane + o < 64) v += u;
    }
    return v;
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// ---- ray / unit-sphere geometry shared by intersect_sphere and depth2pts_outside -------------------
struct Sphere {
    float dd, od, d1, m2, m, len, inv_len, root, d2;
    V3 pmid;
};
__device__ __forceinline__ Sphere sphere_terms(V3 o, V3 d) {
    Sphere s;
    s.dd = dot(d, d);
    s.od = dot(d, o);
    s.d1 = -s.od / s.dd;
    s.pmid = o + s.d1 * d;
    s.m2 = dot(s.pmid, s.pmid);
    s.m = sqrtf(s.m2);
    s.len = sqrtf(s.dd);
    s.inv_len = 1.f / s.len;
    return s;
}
// reverse pass of (d1, pmid, inv_len): accumulates into g_o / g_d given g_d1, g_pmid, g_inv_len
__device__ __forceinline__ void sphere_terms_bwd(V3 o, V3 d, const Sphere& s, float g_d1, V3 g_pmid, float g_inv_len,
                                                 V3* g_o, V3* g_d) {
    // pmid = o + d1 d
    *g_o = *g_o + g_pmid;
    *g_d = *g_d + s.d1 * g_pmid;
    g_d1 += dot(g_pmid, d);
    // inv_len = (d.d)^-1/2
    *g_d = *g_d + (-g_inv_len * s.inv_len * s.inv_len * s.inv_len) * d;
    // d1 = -(d.o) / (d.d)
    *g_o = *g_o + (-g_d1 / s.dd) * d;
    *g_d = *g_d + (-g_d1 / s.dd) * o + (2.f * g_d1 * s.od / (s.dd * s.dd)) * d;
}

// ---- intersect_sphere (ddp_train_nerf.py:50-68) -----------------------------------------------------
__global__ void intersect_fwd_kernel(const float* __restrict__ ro, const float* __restrict__ rd,
                                     float* __restrict__ far, int* __restrict__ outside, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 o = ld3(ro, i), d = ld3(rd, i);
    const Sphere s = sphere_terms(o, d);
    if (s.m2 >= 1.f && outside) *outside = 1;      // the reference raises (:60-64); the host checks (benign race: all write 1)
    far[i] = s.d1 + sqrtf(1.f - s.m2) * s.inv_len;
}

__global__ void intersect_bwd_kernel(const float* __restrict__ ro, const float* __restrict__ rd,
                                     const float* __restrict__ g_far, float* __restrict__ g_o,
                                     float* __restrict__ g_d, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 o = ld3(ro, i), d = ld3(rd, i);
    const Sphere s = sphere_terms(o, d);
    const float g = g_far[i];
    const float root = sqrtf(1.f - s.m2);
    // far = d1 + sqrt(1 - m2) inv_len,  m2 = pmid.pmid
    const float g_m2 = g * s.inv_len * (-0.5f / root);
    V3 go = mk(0, 0, 0), gd = mk(0, 0, 0);
    sphere_terms_bwd(o, d, s, g, (2.f * g_m2) * s.pmid, g * root, &go, &gd);
    st3(g_o, i, go);
    st3(g_d, i, gd);
}

// ---- perturb_samples (ddp_train_nerf.py:71-80) ------------------------------------------------------
__global__ void perturb_fwd_kernel(const float* __restrict__ z, const float* __restrict__ t_rand,
                                   float* __restrict__ out, long n, int s) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n * s) return;
    const int i = (int)(k % s);
    const float zi = z[k];
    const float lower = i > 0 ? 0.5f * (zi + z[k - 1]) : zi;
    const float upper = i + 1 < s ? 0.5f * (z[k + 1] + zi) : zi;
    out[k] = lower + (upper - lower) * t_rand[k];
}

// out_i = lower_i (1 - t_i) + upper_i t_i; lower_i = mid_{i-1} | z_0, upper_i = mid_i | z_{s-1}
__global__ void perturb_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ t_rand,
                                   float* __restrict__ g_z, long n, int s) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n * s) return;
    const int i = (int)(k % s);
    auto gl = [&](long q) { return g_out[q] * (1.f - t_rand[q]); };    // d/d lower_q
    auto gu = [&](long q) { return g_out[q] * t_rand[q]; };            // d/d upper_q
    float g = 0.f;
    // z_i feeds: lower_i (i == 0: whole, else half via mid_{i-1}), upper_i (i == s-1: whole, else half via
    // mid_i), lower_{i+1} (half via mid_i), upper_{i-1} (half via mid_{i-1})
    g += i == 0 ? gl(k) : 0.5f * gl(k);
    g += i == s - 1 ? gu(k) : 0.5f * gu(k);
    if (i + 1 < s) g += 0.5f * gl(k + 1);
    if (i > 0) g += 0.5f * gu(k - 1);
    g_z[k] = g;
}

// ---- sample_pdf, NeRF++ flavour (ddp_train_nerf.py:83-132) ------------------------------------------
// bins [n, m+1], weights [n, m], u [n, ns] -> samples [n, ns]; below / t (optional, for the backward).
__global__ __launch_bounds__(256) void npp_sample_pdf_kernel(
    const float* __restrict__ bins, const float* __restrict__ weights, const float* __restrict__ u,
    float* __restrict__ samples, int* __restrict__ below_out, float* __restrict__ t_out, float* __restrict__ cdf_out,
    int n, int m, int ns, int lds_per_wave) {
    float* s_w = dynamic_lds<float>() + (size_t)wave_id() * lds_per_wave;
    float* s_cdf = s_w + m;                  // m + 1 entries
    const int lane = lane_id();
    int ray = blockIdx.x * kRaysPerBlock + wave_id();
    const bool live = ray < n;
    if (!live) ray = n - 1;
    for (int k = lane; k < m; k += kWave) s_w[k] = weights[(size_t)ray * m + k] + kTiny;
    block_sync();
    if (lane == 0) {
        const float tot = aten_rowsum(s_w, m);
        double run = 0.0;
        s_cdf[0] = 0.f;
        for (int k = 0; k < m; ++k) {
            const float pdf = s_w[k] / tot;
            run += (double)pdf;
            s_cdf[k + 1] = (float)run;
        }
    }
    block_sync();
    if (cdf_out && live)
        for (int k = lane; k <= m; k += kWave) cdf_out[(size_t)ray * (m + 1) + k] = s_cdf[k];
    const float* b = bins + (size_t)ray * (m + 1);
    for (int j = lane; j < ns; j += kWave) {
        const float uq = u[(size_t)ray * ns + j];
        int above = 0;
        for (int k = 0; k < m; ++k) above += (uq >= s_cdf[k]) ? 1 : 0;       // count over cdf[0..m-1] (:113)
        const int below = max(above - 1, 0);
        const float c0 = s_cdf[below], c1 = s_cdf[above];
        float denom = c1 - c0;
        if (denom < kTiny) denom = 1.f;
        const float t = (uq - c0) / denom;
        const float b0 = b[below], b1 = b[above];
        if (live) {
            samples[(size_t)ray * ns + j] = b0 + t * (b1 - b0 + kTiny);
            if (below_out) below_out[(size_t)ray * ns + j] = below | (above << 16);
            if (t_out) t_out[(size_t)ray * ns + j] = t;
        }
    }
}

// d bins: sample = b0 (1 - t) + b1 t + t TINY  (weights are detached by the caller, :457,465)
__global__ __launch_bounds__(256) void npp_sample_pdf_bwd_kernel(
    const float* __restrict__ g_samples, const int* __restrict__ below_above, const float* __restrict__ t,
    float* __restrict__ g_bins, int n, int m, int ns, int lds_per_wave) {
    float* s_g = dynamic_lds<float>() + (size_t)wave_id() * lds_per_wave;      // m + 1
    const int lane = lane_id();
    const int ray = blockIdx.x * kRaysPerBlock + wave_id();
    const bool live = ray < n;
    for (int k = lane; k <= m; k += kWave) s_g[k] = 0.f;
    block_sync();
    if (live) {
        // few collisions per bin; LDS atomics keep it simple (order-dependent rounding only)
        for (int j = lane; j < ns; j += kWave) {
            const float g = g_samples[(size_t)ray * ns + j];
            const int ba = below_above[(size_t)ray * ns + j];
            const float tt = t[(size_t)ray * ns + j];
            atomic_add(&s_g[ba & 0xffff], g * (1.f - tt));
            atomic_add(&s_g[ba >> 16], g * tt);
        }
    }
    block_sync();
    if (live)
        for (int k = lane; k <= m; k += kWave) g_bins[(size_t)ray * (m + 1) + k] = s_g[k];
}

// ---- sample placement (ddp_model.py:80-89 foreground, :16-45 + :105-114 background) -----------------
struct BgRay {
    Sphere s;
    V3 ps, axis_raw, axis;
    float axis_len, phi;
};
__device__ __forceinline__ BgRay bg_ray(V3 o, V3 d) {
    BgRay b;
    b.s = sphere_terms(o, d);
    b.s.root = sqrtf(1.f - b.s.m * b.s.m);
    b.s.d2 = b.s.root * b.s.inv_len;
    b.ps = o + (b.s.d1 + b.s.d2) * d;
    b.axis_raw = cross(o, b.ps);
    b.axis_len = sqrtf(dot(b.axis_raw, b.axis_raw));
    b.axis = mk(b.axis_raw.x / b.axis_len, b.axis_raw.y / b.axis_len, b.axis_raw.z / b.axis_len);
    b.phi = asinf(b.s.m);
    return b;
}
struct BgPoint { float theta, c, s, k, len; V3 axp, rot, unit; };
__device__ __forceinline__ BgPoint bg_point(const BgRay& b, float depth) {
    BgPoint p;
    p.theta = asinf(b.s.m * depth);
    const float ang = b.phi - p.theta;
    p.c = cosf(ang);
    p.s = sinf(ang);
    p.axp = cross(b.axis, b.ps);
    p.k = dot(b.axis, b.ps);
    p.rot = p.c * b.ps + p.s * p.axp + (p.k * (1.f - p.c)) * b.axis;
    p.len = sqrtf(dot(p.rot, p.rot));
    p.unit = mk(p.rot.x / p.len, p.rot.y / p.len, p.rot.z / p.len);
    return p;
}

// fg_pts [n, sf, 3] = o + z d; bg_pts [n, sb, 4] = (unit direction, 1/r) of bg_z in FLIPPED order
// (:113: the network sees them far -> near); viewdirs [n, 3] = d / |d|
__global__ __launch_bounds__(256) void npp_points_fwd_kernel(
    const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ fg_z,
    const float* __restrict__ bg_z, float* __restrict__ fg_pts, float* __restrict__ bg_pts,
    float* __restrict__ viewdirs, float* __restrict__ bg_depth_real, int n, int sf, int sb) {
    const int lane = lane_id();
    const int ray = blockIdx.x * kRaysPerBlock + wave_id();
    if (ray >= n) return;
    const V3 o = ld3(ro, ray), d = ld3(rd, ray);
    for (int i = lane; i < sf; i += kWave) {
        const float z = fg_z[(size_t)ray * sf + i];
        st3(fg_pts, (size_t)ray * sf + i, o + z * d);
    }
    if (sb > 0) {
        const BgRay b = bg_ray(o, d);
        for (int j = lane; j < sb; j += kWave) {
            const float depth = bg_z[(size_t)ray * sb + (sb - 1 - j)];
            const BgPoint p = bg_point(b, depth);
            f32x4 v = {p.unit.x, p.unit.y, p.unit.z, depth};
            *reinterpret_cast<f32x4*>(bg_pts + ((size_t)ray * sb + j) * 4) = v;
            if (bg_depth_real)      // metric depth along the ray (ddp_model.py:44), same flipped order
                bg_depth_real[(size_t)ray * sb + j] = 1.f / (depth + kTiny) * cosf(p.theta) * b.s.inv_len + b.s.d1;
        }
    }
    if (lane == 0 && viewdirs) {
        const float len = sqrtf(dot(d, d));
        st3(viewdirs, ray, mk(d.x / len, d.y / len, d.z / len));
    }
}

// d ray_o, d ray_d, d fg_z from: d fg_pts [n,sf,3], d bg_pts [n,sb,4] (flipped order; the 1/r channel
// carries no gradient back: the inverse radii do not depend on the ray), d viewdirs per sample of both
// networks, d |d| (from the compositing), plus gradients arriving directly (g_z_in: composite's d fg_z).
__global__ __launch_bounds__(256) void npp_points_bwd_kernel(
    const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ fg_z,
    const float* __restrict__ bg_z, const float* __restrict__ d_fg_pts, const float* __restrict__ d_bg_pts,
    const float* __restrict__ d_views_fg, const float* __restrict__ d_views_bg, const float* __restrict__ d_norm,
    const float* __restrict__ g_z_in, float* __restrict__ g_o, float* __restrict__ g_d,
    float* __restrict__ g_fg_z, int n, int sf, int sb) {
    const int lane = lane_id();
    const int ray = blockIdx.x * kRaysPerBlock + wave_id();
    if (ray >= n) return;
    const V3 o = ld3(ro, ray), d = ld3(rd, ray);
    double ao[3] = {0, 0, 0}, ad[3] = {0, 0, 0}, av[3] = {0, 0, 0};
    for (int i = lane; i < sf; i += kWave) {
        const size_t k = (size_t)ray * sf + i;
        const V3 g = ld3(d_fg_pts, k);
        const float z = fg_z[k];
        ao[0] += g.x; ao[1] += g.y; ao[2] += g.z;
        ad[0] += (double)(g.x * z); ad[1] += (double)(g.y * z); ad[2] += (double)(g.z * z);
        g_fg_z[k] = dot(g, d) + (g_z_in ? g_z_in[k] : 0.f);
        const V3 v = ld3(d_views_fg, k);
        av[0] += v.x; av[1] += v.y; av[2] += v.z;
    }
    // background: accumulate the per-ray intermediates' gradients over the samples
    double a_ps[3] = {0, 0, 0}, a_ax[3] = {0, 0, 0}, a_m = 0.0;
    BgRay b;
    if (sb > 0) b = bg_ray(o, d);
    for (int j = lane; j < sb; j += kWave) {
        const size_t k = (size_t)ray * sb + j;
        const float depth = bg_z[(size_t)ray * sb + (sb - 1 - j)];
        const BgPoint p = bg_point(b, depth);
        const V3 g = mk(d_bg_pts[k * 4], d_bg_pts[k * 4 + 1], d_bg_pts[k * 4 + 2]);
        const V3 v = ld3(d_views_bg, k);
        av[0] += v.x; av[1] += v.y; av[2] += v.z;
        // unit = rot / |rot|
        const V3 gr = (1.f / p.len) * (g - dot(p.unit, g) * p.unit);
        // rot = c ps + s (axis x ps) + k (1 - c) axis,  k = axis . ps
        const float ga = dot(gr, b.axis);
        V3 g_ps = p.c * gr + p.s * cross(gr, b.axis) + ((1.f - p.c) * ga) * b.axis;
        V3 g_ax = p.s * cross(b.ps, gr) + (1.f - p.c) * (p.k * gr + ga * b.ps);
        const float g_c = dot(gr, b.ps) - ga * p.k;
        const float g_s = dot(gr, p.axp);
        const float g_ang = -p.s * g_c + p.c * g_s;
        // ang = phi - theta, theta = asin(m depth), phi = asin(m)
        const float md = b.s.m * depth;
        const float g_m = g_ang / sqrtf(1.f - b.s.m * b.s.m) - g_ang * depth / sqrtf(1.f - md * md);
        a_ps[0] += g_ps.x; a_ps[1] += g_ps.y; a_ps[2] += g_ps.z;
        a_ax[0] += g_ax.x; a_ax[1] += g_ax.y; a_ax[2] += g_ax.z;
        a_m += g_m;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ao[c] = wave_sum(ao[c]); ad[c] = wave_sum(ad[c]); av[c] = wave_sum(av[c]);
        a_ps[c] = wave_sum(a_ps[c]); a_ax[c] = wave_sum(a_ax[c]);
    }
    a_m = wave_sum(a_m);
    if (lane != 0) return;
    V3 go = mk((float)ao[0], (float)ao[1], (float)ao[2]);
    V3 gd = mk((float)ad[0], (float)ad[1], (float)ad[2]);
    if (sb > 0) {
        V3 g_ps = mk((float)a_ps[0], (float)a_ps[1], (float)a_ps[2]);
        const V3 g_ax = mk((float)a_ax[0], (float)a_ax[1], (float)a_ax[2]);
        float g_m = (float)a_m;
        // axis = a / |a|, a = o x ps
        const V3 g_a = (1.f / b.axis_len) * (g_ax - dot(b.axis, g_ax) * b.axis);
        go = go + cross(b.ps, g_a);
        g_ps = g_ps + cross(g_a, o);
        // ps = o + (d1 + d2) d
        go = go + g_ps;
        gd = gd + (b.s.d1 + b.s.d2) * g_ps;
        const float g_d12 = dot(g_ps, d);
        // d2 = sqrt(1 - m^2) inv_len
        g_m += g_d12 * (-b.s.m / b.s.root) * b.s.inv_len;
        const float g_inv_len = g_d12 * b.s.root;
        // m = |pmid|
        const V3 g_pmid = (g_m / b.s.m) * b.s.pmid;
        sphere_terms_bwd(o, d, b.s, g_d12, g_pmid, g_inv_len, &go, &gd);
    }
    // viewdirs = d / |d| (ddp_model.py:83) and |d| itself (:93)
    const float len = sqrtf(dot(d, d));
    const V3 v = mk(d.x / len, d.y / len, d.z / len);
    const V3 gv = mk((float)av[0], (float)av[1], (float)av[2]);
    gd = gd + (1.f / len) * (gv - dot(v, gv) * v);
    if (d_norm) gd = gd + d_norm[ray] * v;
    st3(g_o, ray, go);
    st3(g_d, ray, gd);
}

// ---- two-level compositing (ddp_model.py:90-143) ----------------------------------------------------
struct Terms { float sigma, delta, dist, e, alpha, q; };
__device__ __forceinline__ Terms fg_terms(float raw_sigma, float z, float z_next, float norm) {
    Terms t;
    t.sigma = fabsf(raw_sigma);
    t.delta = z_next - z;
    t.dist = norm * t.delta;
    t.e = expf(-t.sigma * t.dist);
    t.alpha = 1.f - t.e;
    t.q = 1.f - t.alpha + kTiny;
    return t;
}

struct CompArgs {
    const float* raw_fg; const float* raw_bg;       // [n,sf,4], [n,sb,4] (bg in flipped order)
    const float* fg_z; const float* fg_z_max; const float* bg_z; const float* rd;
    int n, sf, sb;
};

// one front-to-back sweep over `s` samples: T per sample into s_T (optional), weights out (optional),
// returns the sums needed; foreground when `fg`
template <bool FG>
__device__ __forceinline__ void sweep(const CompArgs& a, int ray, int lane, float norm, float* s_T, float* s_w,
                                      float* w_out, bool live, double* rgb3, double* depth, float* lambda) {
    const int s = FG ? a.sf : a.sb;
    const float* raw = FG ? a.raw_fg : a.raw_bg;
    double carry = 1.0, sr = 0, sg = 0, sb_ = 0, sd = 0;
    for (int base = 0; base < s; base += 64) {
        const int i = base + lane;
        const bool in = i < s;
        const int ic = in ? i : s - 1;
        const f32x4 rw = *reinterpret_cast<const f32x4*>(raw + ((size_t)ray * s + ic) * 4);
        float z, zn;
        if (FG) {
            z = a.fg_z[(size_t)ray * s + ic];
            zn = ic + 1 < s ? a.fg_z[(size_t)ray * s + ic + 1] : a.fg_z_max[ray];
        } else {        // flipped inverse radii run 1 -> 0; interval = this - next, HUGE after the last
            z = a.bg_z[(size_t)ray * s + (s - 1 - ic)];
            zn = ic + 1 < s ? a.bg_z[(size_t)ray * s + (s - 2 - ic)] : 0.f;
        }
        Terms t;
        if (FG) {
            t = fg_terms(rw[3], z, zn, norm);
        } else {
            t.sigma = fabsf(rw[3]);
            t.delta = ic + 1 < s ? z - zn : kHuge;
            t.dist = t.delta;
            t.e = expf(-t.sigma * t.dist);
            t.alpha = 1.f - t.e;
            t.q = 1.f - t.alpha + kTiny;
        }
        const double incl = wave_incl_prod(in ? (double)t.q : 1.0, lane) * carry;
        double excl = shfl_up(incl, 1);
        if (lane == 0) excl = carry;
        carry = shfl(incl, 63);
        const float T = (float)excl;
        const float w = t.alpha * T;
        if (in) {
            if (s_T) s_T[i] = T;
            if (s_w) s_w[i] = w;
            if (live && w_out) w_out[(size_t)ray * s + i] = w;
            sr += (double)(w * sigmoidf(rw[0]));
            sg += (double)(w * sigmoidf(rw[1]));
            sb_ += (double)(w * sigmoidf(rw[2]));
            sd += (double)(w * z);
        }
    }
    rgb3[0] = wave_sum(sr); rgb3[1] = wave_sum(sg); rgb3[2] = wave_sum(sb_);
    *depth = wave_sum(sd);
    if (lambda) *lambda = (float)carry;           // product over ALL samples (:96)
}

__global__ __launch_bounds__(256) void npp_composite_fwd_kernel(
    CompArgs a, float* __restrict__ rgb, float* __restrict__ fg_w, float* __restrict__ bg_w,
    float* __restrict__ fg_rgb, float* __restrict__ fg_depth, float* __restrict__ bg_rgb,
    float* __restrict__ bg_depth, float* __restrict__ bg_lambda) {
    const int lane = lane_id();
    int ray = blockIdx.x * kRaysPerBlock + wave_id();
    const bool live = ray < a.n;
    if (!live) ray = a.n - 1;
    const V3 d = ld3(a.rd, ray);
    const float norm = sqrtf(dot(d, d));
    double f3[3], fd, b3[3], bd;
    float lambda;
    sweep<true>(a, ray, lane, norm, nullptr, nullptr, fg_w, live, f3, &fd, &lambda);
    sweep<false>(a, ray, lane, norm, nullptr, nullptr, bg_w, live, b3, &bd, nullptr);
    if (live && lane == 0) {
        for (int c = 0; c < 3; ++c) {
            const float f = (float)f3[c], b = lambda * (float)b3[c];
            fg_rgb[(size_t)ray * 3 + c] = f;
            bg_rgb[(size_t)ray * 3 + c] = b;
            rgb[(size_t)ray * 3 + c] = f + b;
        }
        fg_depth[ray] = (float)fd;
        bg_depth[ray] = lambda * (float)bd;
        bg_lambda[ray] = lambda;
    }
}

struct CompGrads {
    const float* g_rgb; const float* g_fg_w; const float* g_bg_w; const float* g_fg_rgb; const float* g_fg_depth;
    const float* g_bg_rgb; const float* g_bg_depth; const float* g_lambda;      // any may be NULL
};

// back-to-front sweep: per-sample d raw (+ d delta for the foreground into s_dd)
template <bool FG>
__device__ __forceinline__ double back_sweep(const CompArgs& a, int ray, int lane, float norm, const float* s_T,
                                             float* s_dd, const float G3[3], float Gdepth, const float* g_w,
                                             double suffix0, bool live, float* d_raw) {
    const int s = FG ? a.sf : a.sb;
    const float* raw = FG ? a.raw_fg : a.raw_bg;
    double suffix = suffix0, dnorm = 0.0;
    const int npass = (s + 63) / 64;
    for (int pass = npass - 1; pass >= 0; --pass) {
        const int i = pass * 64 + lane;
        const bool in = i < s;
        const int ic = in ? i : s - 1;
        const f32x4 rw = *reinterpret_cast<const f32x4*>(raw + ((size_t)ray * s + ic) * 4);
        float z, zn;
        Terms t;
        if (FG) {
            z = a.fg_z[(size_t)ray * s + ic];
            zn = ic + 1 < s ? a.fg_z[(size_t)ray * s + ic + 1] : a.fg_z_max[ray];
            t = fg_terms(rw[3], z, zn, norm);
        } else {
            z = a.bg_z[(size_t)ray * s + (s - 1 - ic)];
            zn = ic + 1 < s ? a.bg_z[(size_t)ray * s + (s - 2 - ic)] : 0.f;
            t.sigma = fabsf(rw[3]);
            t.delta = ic + 1 < s ? z - zn : kHuge;
            t.dist = t.delta;
            t.e = expf(-t.sigma * t.dist);
            t.alpha = 1.f - t.e;
            t.q = 1.f - t.alpha + kTiny;
        }
        const float T = s_T[ic];
        const float w = t.alpha * T;
        const float c0 = sigmoidf(rw[0]), c1 = sigmoidf(rw[1]), c2 = sigmoidf(rw[2]);
        const float G = G3[0] * c0 + G3[1] * c1 + G3[2] * c2 + Gdepth * z + (g_w ? g_w[(size_t)ray * s + ic] : 0.f);
        const double gw = in ? (double)G * (double)w : 0.0;
        const double incl = wave_incl_sum_rev(gw, lane) + suffix;
        const double after = incl - gw;
        suffix = shfl(incl, 0);
        if (in) {
            const float dalpha = (float)((double)G * (double)T - after / (double)t.q);
            const float sgn = rw[3] > 0.f ? 1.f : (rw[3] < 0.f ? -1.f : 0.f);      // d|x| (0 at 0, like torch)
            f32x4 o;
            o[0] = G3[0] * w * c0 * (1.f - c0);
            o[1] = G3[1] * w * c1 * (1.f - c1);
            o[2] = G3[2] * w * c2 * (1.f - c2);
            o[3] = sgn * dalpha * t.dist * t.e;
            if (live) *reinterpret_cast<f32x4*>(d_raw + ((size_t)ray * s + i) * 4) = o;
            if (FG) {
                const float ddist = dalpha * t.sigma * t.e;
                s_dd[i] = ddist * norm;                       // d delta_i
                dnorm += (double)(ddist * t.delta);
            }
        }
    }
    return wave_sum(dnorm);
}

__global__ __launch_bounds__(256) void npp_composite_bwd_kernel(
    CompArgs a, CompGrads g, float* __restrict__ d_raw_fg, float* __restrict__ d_raw_bg,
    float* __restrict__ d_fg_z, float* __restrict__ d_z_max, float* __restrict__ d_norm, int lds_per_wave) {
    float* lds = dynamic_lds<float>() + (size_t)wave_id() * lds_per_wave;
    float* s_Tf = lds;                       // sf
    float* s_wf = s_Tf + a.sf;               // sf
    float* s_dd = s_wf + a.sf;               // sf
    float* s_Tb = s_dd + a.sf;               // sb
    const int lane = lane_id();
    int ray = blockIdx.x * kRaysPerBlock + wave_id();
    const bool live = ray < a.n;
    if (!live) ray = a.n - 1;
    const V3 d = ld3(a.rd, ray);
    const float norm = sqrtf(dot(d, d));
    double f3[3], fd, b3[3], bd;
    float lambda;
    sweep<true>(a, ray, lane, norm, s_Tf, s_wf, nullptr, false, f3, &fd, &lambda);
    sweep<false>(a, ray, lane, norm, s_Tb, nullptr, nullptr, false, b3, &bd, nullptr);
    block_sync();
    auto ld = [&](const float* p, int c, int stride) { return p ? p[(size_t)ray * stride + c] : 0.f; };
    float Gf[3], Gb[3], Gb_raw[3];
    float g_lam = ld(g.g_lambda, 0, 1);
    for (int c = 0; c < 3; ++c) {
        const float gr = ld(g.g_rgb, c, 3);
        Gf[c] = gr + ld(g.g_fg_rgb, c, 3);
        Gb[c] = gr + ld(g.g_bg_rgb, c, 3);
        Gb_raw[c] = lambda * Gb[c];
        g_lam += Gb[c] * (float)b3[c];
    }
    const float g_bgd = ld(g.g_bg_depth, 0, 1);
    g_lam += g_bgd * (float)bd;
    const float g_fgd = ld(g.g_fg_depth, 0, 1);
    // background first (independent of lambda's gradient), then the foreground with the lambda term
    back_sweep<false>(a, ray, lane, norm, s_Tb, nullptr, Gb_raw, lambda * g_bgd, g.g_bg_w, 0.0, live, d_raw_bg);
    // lambda = prod_i q_i: d lambda / d alpha_i = -lambda / q_i -- the same form as the "everything behind
    // sample i" term, so it enters as the initial suffix
    const double dn = back_sweep<true>(a, ray, lane, norm, s_Tf, s_dd, Gf, g_fgd, g.g_fg_w,
                                       (double)g_lam * (double)lambda, live, d_raw_fg);
    block_sync();
    if (live) {
        // delta_i = z_{i+1} - z_i (z_max after the last); fg_depth = sum w_i z_i
        for (int i = lane; i < a.sf; i += kWave) {
            const float prev = i > 0 ? s_dd[i - 1] : 0.f;
            d_fg_z[(size_t)ray * a.sf + i] = prev - s_dd[i] + g_fgd * s_wf[i];
        }
        if (lane == 0) {
            d_z_max[ray] = s_dd[a.sf - 1];
            d_norm[ray] = (float)dn;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------- C ABI ----
extern "C" int scnerf_npp_intersect_fwd(const float* ray_o, const float* ray_d, float* far, int* outside_flag,
                                        int n, void* stream) {
    SCN_RETURN_IF(!ray_o || !ray_d || !far || n < 0, SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(intersect_fwd_kernel, dim3(scn_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, ray_o,
                       ray_d, far, outside_flag, n);
    return scn_launch_status();
}

extern "C" int scnerf_npp_intersect_bwd(const float* ray_o, const float* ray_d, const float* g_far, float* g_o,
                                        float* g_d, int n, void* stream) {
    SCN_RETURN_IF(!ray_o || !ray_d || !g_far || !g_o || !g_d || n < 0, SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(intersect_bwd_kernel, dim3(scn_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, ray_o,
                       ray_d, g_far, g_o, g_d, n);
    return scn_launch_status();
}

extern "C" int scnerf_npp_perturb_fwd(const float* z, const float* t_rand, float* out, int n, int s, void* stream) {
    SCN_RETURN_IF(!z || !t_rand || !out || n < 0 || s < 1, SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(perturb_fwd_kernel, dim3(scn_ceil_div((long)n * s, 256)), dim3(256), 0, (hipStream_t)stream,
                       z, t_rand, out, (long)n, s);
    return scn_launch_status();
}

extern "C" int scnerf_npp_perturb_bwd(const float* g_out, const float* t_rand, float* g_z, int n, int s,
                                      void* stream) {
    SCN_RETURN_IF(!g_out || !t_rand || !g_z || n < 0 || s < 1, SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(perturb_bwd_kernel, dim3(scn_ceil_div((long)n * s, 256)), dim3(256), 0, (hipStream_t)stream,
                       g_out, t_rand, g_z, (long)n, s);
    return scn_launch_status();
}

extern "C" int scnerf_npp_sample_pdf(const float* bins, const float* weights, const float* u, float* samples,
                                     int* below_above, float* t, float* cdf, int n, int m, int ns, void* stream) {
    SCN_RETURN_IF(!bins || !weights || !u || !samples || n < 0 || m < 1 || m > 32767 || ns < 1, SCN_EINVAL);
    if (n == 0) return 0;
    const int per_wave = (2 * m + 1 + 3) / 4 * 4;
    const size_t lds = (size_t)per_wave * 4 * kRaysPerBlock;
    SCN_RETURN_IF(lds > 64 * 1024, SCN_ENOSUP);
    hipLaunchKernelGGL(npp_sample_pdf_kernel, dim3(scn_ceil_div(n, kRaysPerBlock)), dim3(256), lds,
                       (hipStream_t)stream, bins, weights, u, samples, below_above, t, cdf, n, m, ns, per_wave);
    return scn_launch_status();
}

extern "C" int scnerf_npp_sample_pdf_bwd(const float* g_samples, const int* below_above, const float* t,
                                         float* g_bins, int n, int m, int ns, void* stream) {
    SCN_RETURN_IF(!g_samples || !below_above || !t || !g_bins || n < 0 || m < 1 || ns < 1, SCN_EINVAL);
    if (n == 0) return 0;
    const int per_wave = (m + 1 + 3) / 4 * 4;
    const size_t lds = (size_t)per_wave * 4 * kRaysPerBlock;
    SCN_RETURN_IF(lds > 64 * 1024, SCN_ENOSUP);
    hipLaunchKernelGGL(npp_sample_pdf_bwd_kernel, dim3(scn_ceil_div(n, kRaysPerBlock)), dim3(256), lds,
                       (hipStream_t)stream, g_samples, below_above, t, g_bins, n, m, ns, per_wave);
    return scn_launch_status();
}

extern "C" int scnerf_npp_points_fwd(const float* ray_o, const float* ray_d, const float* fg_z, const float* bg_z,
                                     float* fg_pts, float* bg_pts, float* viewdirs, float* bg_depth_real, int n,
                                     int sf, int sb, void* stream) {
    SCN_RETURN_IF(!ray_o || !ray_d || n < 0 || sf < 0 || sb < 0 || (sf > 0 && (!fg_z || !fg_pts)), SCN_EINVAL);
    SCN_RETURN_IF(sb > 0 && (!bg_z || !bg_pts), SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(npp_points_fwd_kernel, dim3(scn_ceil_div(n, kRaysPerBlock)), dim3(256), 0,
                       (hipStream_t)stream, ray_o, ray_d, fg_z, bg_z, fg_pts, bg_pts, viewdirs, bg_depth_real, n, sf, sb);
    return scn_launch_status();
}

extern "C" int scnerf_npp_points_bwd(const float* ray_o, const float* ray_d, const float* fg_z, const float* bg_z,
                                     const float* d_fg_pts, const float* d_bg_pts, const float* d_views_fg,
                                     const float* d_views_bg, const float* d_norm, const float* g_fg_z_in,
                                     float* g_ray_o, float* g_ray_d, float* g_fg_z, int n, int sf, int sb,
                                     void* stream) {
    SCN_RETURN_IF(!ray_o || !ray_d || !fg_z || !d_fg_pts || !d_views_fg || !g_ray_o || !g_ray_d || !g_fg_z, SCN_EINVAL);
    SCN_RETURN_IF(n < 0 || sf < 1 || sb < 0 || (sb > 0 && (!bg_z || !d_bg_pts || !d_views_bg)), SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(npp_points_bwd_kernel, dim3(scn_ceil_div(n, kRaysPerBlock)), dim3(256), 0,
                       (hipStream_t)stream, ray_o, ray_d, fg_z, bg_z, d_fg_pts, d_bg_pts, d_views_fg, d_views_bg,
                       d_norm, g_fg_z_in, g_ray_o, g_ray_d, g_fg_z, n, sf, sb);
    return scn_launch_status();
}

extern "C" int scnerf_npp_composite_fwd(const float* raw_fg, const float* raw_bg, const float* fg_z,
                                        const float* fg_z_max, const float* bg_z, const float* ray_d, float* rgb,
                                        float* fg_weights, float* bg_weights, float* fg_rgb, float* fg_depth,
                                        float* bg_rgb, float* bg_depth, float* bg_lambda, int n, int sf, int sb,
                                        void* stream) {
    SCN_RETURN_IF(!raw_fg || !raw_bg || !fg_z || !fg_z_max || !bg_z || !ray_d || !rgb || !fg_rgb || !fg_depth, SCN_EINVAL);
    SCN_RETURN_IF(!bg_rgb || !bg_depth || !bg_lambda || n < 0 || sf < 1 || sb < 1, SCN_EINVAL);
    if (n == 0) return 0;
    CompArgs a;
    a.raw_fg = raw_fg; a.raw_bg = raw_bg; a.fg_z = fg_z; a.fg_z_max = fg_z_max; a.bg_z = bg_z; a.rd = ray_d;
    a.n = n; a.sf = sf; a.sb = sb;
    hipLaunchKernelGGL(npp_composite_fwd_kernel, dim3(scn_ceil_div(n, kRaysPerBlock)), dim3(256), 0,
                       (hipStream_t)stream, a, rgb, fg_weights, bg_weights, fg_rgb, fg_depth, bg_rgb, bg_depth,
                       bg_lambda);
    return scn_launch_status();
}

extern "C" int scnerf_npp_composite_bwd(const float* raw_fg, const float* raw_bg, const float* fg_z,
                                        const float* fg_z_max, const float* bg_z, const float* ray_d,
                                        const float* g_rgb, const float* g_fg_weights, const float* g_bg_weights,
                                        const float* g_fg_rgb, const float* g_fg_depth, const float* g_bg_rgb,
                                        const float* g_bg_depth, const float* g_bg_lambda, float* d_raw_fg,
                                        float* d_raw_bg, float* d_fg_z, float* d_fg_z_max, float* d_norm, int n,
                                        int sf, int sb, void* stream) {
    SCN_RETURN_IF(!raw_fg || !raw_bg || !fg_z || !fg_z_max || !bg_z || !ray_d, SCN_EINVAL);
    SCN_RETURN_IF(!d_raw_fg || !d_raw_bg || !d_fg_z || !d_fg_z_max || !d_norm || n < 0 || sf < 1 || sb < 1, SCN_EINVAL);
    if (n == 0) return 0;
    CompArgs a;
    a.raw_fg = raw_fg; a.raw_bg = raw_bg; a.fg_z = fg_z; a.fg_z_max = fg_z_max; a.bg_z = bg_z; a.rd = ray_d;
    a.n = n; a.sf = sf; a.sb = sb;
    CompGrads g;
    g.g_rgb = g_rgb; g.g_fg_w = g_fg_weights; g.g_bg_w = g_bg_weights; g.g_fg_rgb = g_fg_rgb;
    g.g_fg_depth = g_fg_depth; g.g_bg_rgb = g_bg_rgb; g.g_bg_depth = g_bg_depth; g.g_lambda = g_bg_lambda;
    const int per_wave = (3 * sf + sb + 3) / 4 * 4;
    const size_t lds = (size_t)per_wave * 4 * kRaysPerBlock;
    SCN_RETURN_IF(lds > 64 * 1024, SCN_ENOSUP);
    hipLaunchKernelGGL(npp_composite_bwd_kernel, dim3(scn_ceil_div(n, kRaysPerBlock)), dim3(256), lds,
                       (hipStream_t)stream, a, g, d_raw_fg, d_raw_bg, d_fg_z, d_fg_z_max, d_norm,
                       per_wave);
    return scn_launch_status();
}
