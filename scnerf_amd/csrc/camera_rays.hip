// camera_rays.hip -- differentiable camera ray generator for gfx950.
//
// One thread per ray does what the reference spreads over ~40 tensor ops and two full-image
// F.interpolate calls per step:
//   K^-1 from the learnable intrinsics, 6-D (Gram-Schmidt) rotation + translation with learnable
//   residuals, pixel -> direction, bilinear lookup of the ray-origin / ray-direction noise grids at
//   the truncated pixel, renormalisation.
// Replaces /root/reference NeRF/get_rays.py:5-148, model/camera_model.py:24-46, :166-190,
// model/camera_utils.py:78-133, :191-195 and the NDC warps of NeRF/render.py:357-396.
// HBM-bound, tiny: the point is one launch instead of dozens, and direct 4-tap sampling of the
// 37x50 grids instead of materialising two HxWx3 images.
#include <scn_wave.h>

#include "launch.h"
#include "scnerf_hip.h"

namespace {

using namespace scn;

struct Vec3 { float x, y, z; };
__device__ __forceinline__ Vec3 v3(float x, float y, float z) { Vec3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ Vec3 operator+(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ Vec3 operator-(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ Vec3 operator*(float s, Vec3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ Vec3 cross(Vec3 a, Vec3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

struct Rot { Vec3 x, y, z; };   // columns of R

// ortho2rotation (model/camera_utils.py:78-133), including its clamp/eps conventions
struct GramSchmidt {
    Vec3 a1, a2, x, v, y;
    float n1, n1c, dotxa, nx2, nx2c, f, nv, nvc;
};

__device__ __forceinline__ Rot gram_schmidt(const float* p6, GramSchmidt* s) {
    s->a1 = v3(p6[0], p6[1], p6[2]);
    s->a2 = v3(p6[3], p6[4], p6[5]);
    s->n1 = sqrtf(dot(s->a1, s->a1));
    s->n1c = fmaxf(s->n1, 1e-8f);
    s->x = v3(s->a1.x / (s->n1c + 1e-10f), s->a1.y / (s->n1c + 1e-10f), s->a1.z / (s->n1c + 1e-10f));
    s->dotxa = dot(s->x, s->a2);
    s->nx2 = dot(s->x, s->x);
    s->nx2c = fmaxf(s->nx2, 1e-8f);
    s->f = s->dotxa / (s->nx2c + 1e-10f);
    s->v = s->a2 - s->f * s->x;
    s->nv = sqrtf(dot(s->v, s->v));
    s->nvc = fmaxf(s->nv, 1e-8f);
    s->y = v3(s->v.x / (s->nvc + 1e-10f), s->v.y / (s->nvc + 1e-10f), s->v.z / (s->nvc + 1e-10f));
    Rot r;
    r.x = s->x;
    r.y = s->y;
    r.z = cross(s->x, s->y);
    return r;
}

// reverse mode of gram_schmidt: gradients of the three columns -> gradient of the 6 parameters
__device__ __forceinline__ void gram_schmidt_bwd(const GramSchmidt& s, Vec3 gx, Vec3 gy, Vec3 gz, float* g6) {
    gx = gx + cross(s.y, gz);
    gy = gy + cross(gz, s.x);
    const float dv = s.nvc + 1e-10f;
    Vec3 gv = (1.f / dv) * gy;
    if (s.nv > 1e-8f) gv = gv - (dot(gy, s.v) / (dv * dv * s.nv)) * s.v;
    Vec3 ga2 = gv;
    const float gf = -dot(gv, s.x);
    gx = gx - s.f * gv;
    const float dn = s.nx2c + 1e-10f;
    const float gdot = gf / dn;
    const float gnx2 = s.nx2 > 1e-8f ? -gf * s.dotxa / (dn * dn) : 0.f;
    gx = gx + gdot * s.a2 + (2.f * gnx2) * s.x;
    ga2 = ga2 + gdot * s.x;
    const float d1 = s.n1c + 1e-10f;
    Vec3 ga1 = (1.f / d1) * gx;
    if (s.n1 > 1e-8f) ga1 = ga1 - (dot(gx, s.a1) / (d1 * d1 * s.n1)) * s.a1;
    g6[0] = ga1.x; g6[1] = ga1.y; g6[2] = ga1.z;
    g6[3] = ga2.x; g6[4] = ga2.y; g6[5] = ga2.z;
}

// bilinear taps of F.interpolate(..., (H, W), mode='bilinear', align_corners=False) at output pixel
// (py, px) of a [gh, gw, 3] grid (ATen upsample_bilinear2d: src = scale * (dst + .5) - .5, >= 0)
struct Taps { int y0, y1, x0, x1; float wy0, wy1, wx0, wx1; };

__device__ __forceinline__ Taps bilinear_taps(int py, int px, int gh, int gw, int H, int W) {
    Taps t;
    const float sy = (float)gh / (float)H, sx = (float)gw / (float)W;
    // ATen's CPU kernel is built with FP contraction, so its source index is one fused multiply-add
    float fy = fmaf(sy, (float)py + 0.5f, -0.5f);
    float fx = fmaf(sx, (float)px + 0.5f, -0.5f);
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    t.y0 = (int)fy;
    t.x0 = (int)fx;
    t.y1 = t.y0 + (t.y0 < gh - 1 ? 1 : 0);
    t.x1 = t.x0 + (t.x0 < gw - 1 ? 1 : 0);
    t.wy1 = fy - (float)t.y0;
    t.wx1 = fx - (float)t.x0;
    t.wy0 = 1.f - t.wy1;
    t.wx0 = 1.f - t.wx1;
    return t;
}

__device__ __forceinline__ Vec3 grid_at(const float* g, int gw, int y, int x) {
    const float* p = g + ((size_t)y * gw + x) * 3;
    return v3(p[0], p[1], p[2]);
}

__device__ __forceinline__ Vec3 sample_grid(const float* g, int gw, const Taps& t) {
    const Vec3 a = t.wx0 * grid_at(g, gw, t.y0, t.x0) + t.wx1 * grid_at(g, gw, t.y0, t.x1);
    const Vec3 b = t.wx0 * grid_at(g, gw, t.y1, t.x0) + t.wx1 * grid_at(g, gw, t.y1, t.x1);
    return t.wy0 * a + t.wy1 * b;
}

__device__ __forceinline__ void scatter_grid(float* dg, int gw, const Taps& t, Vec3 g) {
    const float w[4] = {t.wy0 * t.wx0, t.wy0 * t.wx1, t.wy1 * t.wx0, t.wy1 * t.wx1};
    const int ys[4] = {t.y0, t.y0, t.y1, t.y1}, xs[4] = {t.x0, t.x1, t.x0, t.x1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float* p = dg + ((size_t)ys[k] * gw + xs[k]) * 3;
        atomic_add(p + 0, w[k] * g.x);
        atomic_add(p + 1, w[k] * g.y);
        atomic_add(p + 2, w[k] * g.z);
    }
}

struct CamArgs {
    const float* kps;            // [n,2] (x, y) or NULL: every pixel of the H x W image
    const long long* cam_idx;    // [n] or NULL
    int single_idx;              // camera used when cam_idx == NULL and extrinsic == NULL
    const float* extrinsic;      // explicit [n_ext,4,4] (n_ext in {1, n}) or NULL
    int n_ext;
    const float* intr_init; const float* intr_noise; float intr_scale; int multiplicative;
    const float* extr_init; const float* extr_noise; float extr_scale; int n_cams;
    const float* grid_o; float scale_o; const float* grid_d; float scale_d; int gh, gw;
    int H, W, n;
    // NeRF++ generator (render_ray_from_camera, nerfplusplus/nerf_sample_ray_split.py:196-257):
    const long long* select;     // non-NULL selects it: row-major pixel indices, rays through pixel CENTRES
    const float* dist;           // [2] radial distortion (k0, k1) or NULL
};

struct Intrinsics { float fx, fy, cx, cy; };

__device__ __forceinline__ Intrinsics intrinsics_of(const CamArgs& a) {
    float p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float init = a.intr_init[i];
        p[i] = a.multiplicative ? init + a.intr_noise[i] * a.intr_scale * init
                                : init + a.intr_noise[i] * a.intr_scale;
    }
    Intrinsics k;
    k.fx = p[0]; k.fy = p[1]; k.cx = p[2]; k.cy = p[3];
    return k;
}

struct RayFwd {
    Vec3 dirs, rd_raw, t;
    Rot R;
    float nrm;
    Taps taps;
    float x, y;
    int cam;
    GramSchmidt gs;
    float ux, uy, rx, ry, sx, sy;      // NeRF++ distortion: offsets from the centre, r = u / c, scale 1 + r^2 k0 + r^4 k1
};

__device__ __forceinline__ void ray_forward(const CamArgs& a, int i, const Intrinsics& K, RayFwd* f, Vec3* ro,
                                            Vec3* rd) {
    float x, y;
    int px = 0, py = 0;                  // pixel whose ray-noise sample is used
    const float ia = 1.f / K.fx, ic = -K.cx / K.fx, ib = 1.f / K.fy, id = -K.cy / K.fy;
    if (a.select) {
        const long long sel = a.select[i];
        px = (int)(sel % a.W); py = (int)(sel / a.W);
        x = (float)px + 0.5f; y = (float)py + 0.5f;                    // pixel centres (:220-221)
        if (a.dist) {                                                   // (:225-232)
            const float k0 = a.dist[0], k1 = a.dist[1];
            f->ux = x - K.cx; f->uy = y - K.cy;
            f->rx = f->ux / K.cx; f->ry = f->uy / K.cy;
            const float rx2 = f->rx * f->rx, ry2 = f->ry * f->ry;
            f->sx = 1.f + rx2 * k0 + rx2 * rx2 * k1;
            f->sy = 1.f + ry2 * k0 + ry2 * ry2 * k1;
            x = f->ux * f->sx + K.cx;
            y = f->uy * f->sy + K.cy;
        }
        f->x = x; f->y = y;
        f->dirs = v3(x * ia + ic, y * ib + id, 1.f);                    // K^-1 [u, v, 1], no axis flips (:234-243)
    } else {
        if (a.kps) { x = a.kps[(size_t)i * 2]; y = a.kps[(size_t)i * 2 + 1]; }
        else { x = (float)(i % a.W); y = (float)(i / a.W); }
        px = (int)x; py = (int)y;                                       // .long() truncation
        f->x = x; f->y = y;
        // dirs = [x, y, 1] K^-T with K^-1 = [[1/fx, 0, -cx/fx], [0, 1/fy, -cy/fy], [0, 0, 1]]  (get_rays.py:119-125)
        f->dirs = v3(x * ia + ic, -(y * ib + id), -1.f);
    }
    if (a.extrinsic) {
        const float* E = a.extrinsic + (size_t)(a.n_ext == 1 ? 0 : i) * 16;
        f->R.x = v3(E[0], E[4], E[8]);
        f->R.y = v3(E[1], E[5], E[9]);
        f->R.z = v3(E[2], E[6], E[10]);
        f->t = v3(E[3], E[7], E[11]);
        f->cam = -1;
    } else {
        const int c = a.cam_idx ? (int)a.cam_idx[i] : a.single_idx;
        f->cam = c;
        float p9[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) p9[k] = a.extr_init[(size_t)c * 9 + k] + a.extr_scale * a.extr_noise[(size_t)c * 9 + k];
        f->R = gram_schmidt(p9, &f->gs);
        f->t = v3(p9[6], p9[7], p9[8]);
    }
    // rays_d[k] = sum_j dirs[j] R[k][j]
    const Vec3 d = f->dirs;
    Vec3 r = v3(d.x * f->R.x.x + d.y * f->R.y.x + d.z * f->R.z.x,
                d.x * f->R.x.y + d.y * f->R.y.y + d.z * f->R.z.y,
                d.x * f->R.x.z + d.y * f->R.y.z + d.z * f->R.z.z);
    Vec3 o = f->t;
    // the pixel the noise grids are sampled at is kept inside the image (callers validate the key points, but
    // the host layer's check is asynchronous: an out-of-range batch must not read or scatter out of bounds)
    px = min(max(px, 0), a.W - 1);
    py = min(max(py, 0), a.H - 1);
    if (a.grid_o || a.grid_d) f->taps = bilinear_taps(py, px, a.gh, a.gw, a.H, a.W);
    if (a.grid_o) o = o + a.scale_o * sample_grid(a.grid_o, a.gw, f->taps);
    if (a.grid_d) {
        r = r + a.scale_d * sample_grid(a.grid_d, a.gw, f->taps);
        f->rd_raw = r;
        f->nrm = sqrtf(dot(r, r));
        const float den = a.select ? f->nrm : f->nrm + 1e-10f;        // NeRF++ renormalises without an epsilon (:254)
        r = v3(r.x / den, r.y / den, r.z / den);
    }
    *ro = o;
    *rd = r;
}

__global__ __launch_bounds__(256) void camera_rays_fwd_kernel(CamArgs a, float* __restrict__ rays_o,
                                                              float* __restrict__ rays_d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const Intrinsics K = intrinsics_of(a);
    RayFwd f;
    Vec3 o, d;
    ray_forward(a, i, K, &f, &o, &d);
    rays_o[(size_t)i * 3 + 0] = o.x; rays_o[(size_t)i * 3 + 1] = o.y; rays_o[(size_t)i * 3 + 2] = o.z;
    rays_d[(size_t)i * 3 + 0] = d.x; rays_d[(size_t)i * 3 + 1] = d.y; rays_d[(size_t)i * 3 + 2] = d.z;
}

// Per-ray reverse pass.  acc layout (floats): [0..3] d intrinsic params (fx, fy, cx, cy);
// then per camera (or per explicit matrix) 12 floats: dR columns x, y, z (9) + dt (3).
// lds_slots > 0: the 12 pose accumulators of each of `lds_slots` cameras are first summed in LDS by the workgroup
// and flushed with one global atomic per non-zero entry -- thousands of rays share a few dozen cameras, and
// 12 global float atomics per ray on ~200 addresses serialised the kernel (94 us at 4096 rays, 17 views)
__global__ __launch_bounds__(256) void camera_rays_bwd_kernel(CamArgs a, const float* __restrict__ g_o,
                                                              const float* __restrict__ g_d, float* acc,
                                                              float* d_grid_o, float* d_grid_d, float* acc_dist,
                                                              int lds_slots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float* lacc = dynamic_lds<float>();
    if (lds_slots > 0) {
        for (int k = threadIdx.x; k < lds_slots * 12; k += blockDim.x) lacc[k] = 0.f;
        block_sync();
    }
    float gk[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // fx, fy, cx, cy, (k0, k1)
    if (i < a.n) {
        const Intrinsics K = intrinsics_of(a);
        RayFwd f;
        Vec3 o, d;
        ray_forward(a, i, K, &f, &o, &d);
        const Vec3 go = g_o ? v3(g_o[(size_t)i * 3], g_o[(size_t)i * 3 + 1], g_o[(size_t)i * 3 + 2]) : v3(0, 0, 0);
        Vec3 gr = g_d ? v3(g_d[(size_t)i * 3], g_d[(size_t)i * 3 + 1], g_d[(size_t)i * 3 + 2]) : v3(0, 0, 0);
        if (a.grid_d) {
            const float den = a.select ? f.nrm : f.nrm + 1e-10f;
            Vec3 g = (1.f / den) * gr;
            if (f.nrm > 0.f) g = g - (dot(gr, f.rd_raw) / (den * den * f.nrm)) * f.rd_raw;
            gr = g;
            if (d_grid_d) scatter_grid(d_grid_d, a.gw, f.taps, a.scale_d * gr);
        }
        if (a.grid_o && d_grid_o) scatter_grid(d_grid_o, a.gw, f.taps, a.scale_o * go);
        // R, t
        const int slot = a.extrinsic ? (a.n_ext == 1 ? 0 : i) : f.cam;
        const Vec3 dd = f.dirs;
        const float term[12] = {gr.x * dd.x, gr.y * dd.x, gr.z * dd.x,        // d col x
                                gr.x * dd.y, gr.y * dd.y, gr.z * dd.y,        // d col y
                                gr.x * dd.z, gr.y * dd.z, gr.z * dd.z,        // d col z
                                go.x, go.y, go.z};
        // Two branches, not one pointer chosen between LDS and global memory: through a pointer of unknown address space the adds
        // become FLAT atomics aimed at LDS (flat_atomic_add_f32, the only ones in the library); this way they are ds_add_f32.
        if (lds_slots > 0) {
            float* ar = lacc + slot * 12;
#pragma unroll
            for (int k = 0; k < 12; ++k) atomic_add(ar + k, term[k]);
        } else {
            float* ar = acc + 4 + (size_t)slot * 12;
#pragma unroll
            for (int k = 0; k < 12; ++k) atomic_add(ar + k, term[k]);
        }
        // dirs -> intrinsics
        const float gdx = dot(gr, f.R.x), gdy = dot(gr, f.R.y);
        if (!a.select) {
            gk[0] = gdx * (-(f.x - K.cx) / (K.fx * K.fx));
            gk[2] = gdx * (-1.f / K.fx);
            gk[1] = gdy * ((f.y - K.cy) / (K.fy * K.fy));
            gk[3] = gdy * (1.f / K.fy);
        } else {
            // dirs_x = x' / fx - cx / fx,  dirs_y = y' / fy - cy / fy  (no sign flip)
            gk[0] = gdx * (-(f.x - K.cx) / (K.fx * K.fx));
            gk[1] = gdy * (-(f.y - K.cy) / (K.fy * K.fy));
            float dxdc = 0.f, dydc = 0.f;                // d x' / d cx, d y' / d cy
            if (a.dist) {
                // x' = u s + c, u = p - c, r = u / c, s = 1 + r^2 k0 + r^4 k1
                const float k0 = a.dist[0], k1 = a.dist[1];
                const float px = f.ux + K.cx, py = f.uy + K.cy;          // the undistorted pixel centre
                const float dsx = (2.f * f.rx * k0 + 4.f * f.rx * f.rx * f.rx * k1) * (-px / (K.cx * K.cx));
                const float dsy = (2.f * f.ry * k0 + 4.f * f.ry * f.ry * f.ry * k1) * (-py / (K.cy * K.cy));
                dxdc = -f.sx + f.ux * dsx + 1.f;
                dydc = -f.sy + f.uy * dsy + 1.f;
                const float gx = gdx / K.fx, gy = gdy / K.fy;            // d L / d x', d L / d y'
                gk[4] = gx * f.ux * f.rx * f.rx + gy * f.uy * f.ry * f.ry;
                gk[5] = gx * f.ux * f.rx * f.rx * f.rx * f.rx + gy * f.uy * f.ry * f.ry * f.ry * f.ry;
            }
            gk[2] = gdx * ((dxdc - 1.f) / K.fx);
            gk[3] = gdy * ((dydc - 1.f) / K.fy);
        }
    }
    if (lds_slots > 0) {
        block_sync();
        for (int k = threadIdx.x; k < lds_slots * 12; k += blockDim.x) {
            const float v = lacc[k];
            if (v != 0.f) atomic_add(acc + 4 + k, v);
        }
    }
    // wave reduce, then one atomic per wave
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float v = gk[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += shfl_xor(v, o);
        if (lane_id() == 0 && (k < 4 || acc_dist)) atomic_add(k < 4 ? acc + k : acc_dist + (k - 4), v);
    }
}

// accumulators -> parameter gradients
__global__ void camera_finish_kernel(CamArgs a, const float* __restrict__ acc, float* __restrict__ d_intr_noise,
                                     float* __restrict__ d_extr_noise, float* __restrict__ d_extrinsic) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && d_intr_noise) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            d_intr_noise[i] = acc[i] * a.intr_scale * (a.multiplicative ? a.intr_init[i] : 1.f);
    }
    if (a.extrinsic) {
        if (d_extrinsic && c < a.n_ext) {
            const float* ar = acc + 4 + (size_t)c * 12;
            float* E = d_extrinsic + (size_t)c * 16;
            // E[k][j] = R[k][j] (column j is accumulator block j), E[k][3] = t[k]
            for (int k = 0; k < 3; ++k) {
                E[k * 4 + 0] = ar[0 + k]; E[k * 4 + 1] = ar[3 + k]; E[k * 4 + 2] = ar[6 + k]; E[k * 4 + 3] = ar[9 + k];
            }
            E[12] = E[13] = E[14] = E[15] = 0.f;
        }
        return;
    }
    if (c < a.n_cams && d_extr_noise) {
        const float* ar = acc + 4 + (size_t)c * 12;
        float p9[9];
        for (int k = 0; k < 9; ++k) p9[k] = a.extr_init[(size_t)c * 9 + k] + a.extr_scale * a.extr_noise[(size_t)c * 9 + k];
        GramSchmidt s;
        gram_schmidt(p9, &s);
        float g6[6];
        gram_schmidt_bwd(s, v3(ar[0], ar[1], ar[2]), v3(ar[3], ar[4], ar[5]), v3(ar[6], ar[7], ar[8]), g6);
        for (int k = 0; k < 6; ++k) d_extr_noise[(size_t)c * 9 + k] = a.extr_scale * g6[k];
        for (int k = 0; k < 3; ++k) d_extr_noise[(size_t)c * 9 + 6 + k] = a.extr_scale * ar[9 + k];
    }
}

__global__ __launch_bounds__(256) void pinhole_rays_kernel(const float* __restrict__ kps, int kps_stride,
                                                           const float* __restrict__ c2w, float focal, int H, int W,
                                                           float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                           int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x, y;
    if (kps) { x = (float)(long long)kps[(size_t)i * kps_stride]; y = (float)(long long)kps[(size_t)i * kps_stride + 1]; }
    else { x = (float)(i % W); y = (float)(i / W); }
    const float dx = (x - (float)W * .5f) / focal, dy = -(y - (float)H * .5f) / focal, dz = -1.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rays_d[(size_t)i * 3 + k] = dx * c2w[k * 4 + 0] + dy * c2w[k * 4 + 1] + dz * c2w[k * 4 + 2];
        rays_o[(size_t)i * 3 + k] = c2w[k * 4 + 3];
    }
}

// ---- NDC warp (render.py:357-396) -------------------------------------------------------------
struct Ndc { float t, ox, oy, oz, sx, sy; };

__device__ __forceinline__ Ndc ndc_forward(int H, int W, float fx, float fy, float near, const float* o,
                                           const float* d, float* no, float* nd) {
    Ndc s;
    s.t = -(near + o[2]) / d[2];
    s.ox = o[0] + s.t * d[0];
    s.oy = o[1] + s.t * d[1];
    s.oz = o[2] + s.t * d[2];
    s.sx = -1.f / ((float)W / (2.f * fx));
    s.sy = -1.f / ((float)H / (2.f * fy));
    no[0] = s.sx * s.ox / s.oz;
    no[1] = s.sy * s.oy / s.oz;
    no[2] = 1.f + 2.f * near / s.oz;
    nd[0] = s.sx * (d[0] / d[2] - s.ox / s.oz);
    nd[1] = s.sy * (d[1] / d[2] - s.oy / s.oz);
    nd[2] = -2.f * near / s.oz;
    return s;
}

__global__ __launch_bounds__(256) void ndc_fwd_kernel(int H, int W, const float* __restrict__ f2, float near,
                                                      const float* __restrict__ o, const float* __restrict__ d,
                                                      float* __restrict__ no, float* __restrict__ nd, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ndc_forward(H, W, f2[0], f2[1], near, o + (size_t)i * 3, d + (size_t)i * 3, no + (size_t)i * 3, nd + (size_t)i * 3);
}

// gradient of the NDC warp of one ray: (a = d L / d ndc_o, b = d L / d ndc_d) -> g_o, g_d, and the ray's
// contributions to d L / d fx, d L / d fy
__device__ __forceinline__ void ndc_backward(int H, int W, float fx, float fy, float near, const float* oi, const float* di,
                                             const float* a, const float* b, float* g_o, float* g_d, float* gfx, float* gfy) {
    float no[3], nd[3];
    const Ndc s = ndc_forward(H, W, fx, fy, near, oi, di, no, nd);
    const float a0 = a[0], a1 = a[1], a2 = a[2], b0 = b[0], b1 = b[1], b2 = b[2];
    const float A = s.ox / s.oz, B = s.oy / s.oz;
    const float g_sx = a0 * A + b0 * (di[0] / di[2] - A);
    const float g_sy = a1 * B + b1 * (di[1] / di[2] - B);
    *gfx = g_sx * (-2.f / (float)W);
    *gfy = g_sy * (-2.f / (float)H);
    const float gA = (a0 - b0) * s.sx, gB = (a1 - b1) * s.sy;
    const float gox = gA / s.oz, goy = gB / s.oz;
    const float goz = -(gA * A + gB * B) / s.oz + (b2 - a2) * (2.f * near / (s.oz * s.oz));
    float gdx = b0 * s.sx / di[2], gdy = b1 * s.sy / di[2];
    float gdz = -(b0 * s.sx * di[0] + b1 * s.sy * di[1]) / (di[2] * di[2]);
    const float gt = gox * di[0] + goy * di[1] + goz * di[2];
    gdx += s.t * gox; gdy += s.t * goy; gdz += s.t * goz;
    const float gz_o = goz + gt * (-1.f / di[2]);
    gdz += gt * (near + oi[2]) / (di[2] * di[2]);
    g_o[0] = gox; g_o[1] = goy; g_o[2] = gz_o;
    g_d[0] = gdx; g_d[1] = gdy; g_d[2] = gdz;
}

__global__ __launch_bounds__(256) void ndc_bwd_kernel(int H, int W, const float* __restrict__ f2, float near,
                                                      const float* __restrict__ o, const float* __restrict__ d,
                                                      const float* __restrict__ g_no, const float* __restrict__ g_nd,
                                                      float* __restrict__ g_o, float* __restrict__ g_d,
                                                      float* g_f2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float gfx = 0.f, gfy = 0.f;
    if (i < n) {
        float a[3] = {0.f, 0.f, 0.f}, b[3] = {0.f, 0.f, 0.f};
        if (g_no) { a[0] = g_no[(size_t)i * 3]; a[1] = g_no[(size_t)i * 3 + 1]; a[2] = g_no[(size_t)i * 3 + 2]; }
        if (g_nd) { b[0] = g_nd[(size_t)i * 3]; b[1] = g_nd[(size_t)i * 3 + 1]; b[2] = g_nd[(size_t)i * 3 + 2]; }
        ndc_backward(H, W, f2[0], f2[1], near, o + (size_t)i * 3, d + (size_t)i * 3, a, b, g_o + (size_t)i * 3,
                     g_d + (size_t)i * 3, &gfx, &gfy);
    }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) { gfx += shfl_xor(gfx, o2); gfy += shfl_xor(gfy, o2); }
    if (g_f2 && lane_id() == 0) { atomic_add(g_f2, gfx); atomic_add(g_f2 + 1, gfy); }
}

// ---- ray-batch packing: what render() does between the ray source and batchify_rays (render.py:105-128) ----
// row = [o' (3), d' (3), near, far (, viewdir (3))]: the view direction is d / |d| of the UN-warped ray (:105-109),
// (o', d') the NDC warp when `f2` is given (ndc = True), else (o, d); one thread per ray instead of ~10
// element-wise launches (and as many again in the backward).
__global__ __launch_bounds__(256) void pack_rays_fwd_kernel(int H, int W, const float* __restrict__ f2, float ndc_near,
                                                            const float* __restrict__ o, const float* __restrict__ d,
                                                            float near, float far, int cols, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* oi = o + (size_t)i * 3;
    const float* di = d + (size_t)i * 3;
    float* r = out + (size_t)i * cols;
    if (cols > 8) {
        const float nrm = sqrtf(di[0] * di[0] + di[1] * di[1] + di[2] * di[2]);
        r[8] = di[0] / nrm; r[9] = di[1] / nrm; r[10] = di[2] / nrm;
    }
    if (f2) {
        float no[3], nd[3];
        ndc_forward(H, W, f2[0], f2[1], ndc_near, oi, di, no, nd);
        r[0] = no[0]; r[1] = no[1]; r[2] = no[2]; r[3] = nd[0]; r[4] = nd[1]; r[5] = nd[2];
    } else {
        r[0] = oi[0]; r[1] = oi[1]; r[2] = oi[2]; r[3] = di[0]; r[4] = di[1]; r[5] = di[2];
    }
    r[6] = near; r[7] = far;
}

__global__ __launch_bounds__(256) void pack_rays_bwd_kernel(int H, int W, const float* __restrict__ f2, float ndc_near,
                                                            const float* __restrict__ o, const float* __restrict__ d,
                                                            int cols, const float* __restrict__ g, float* __restrict__ g_o,
                                                            float* __restrict__ g_d, float* g_f2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float gfx = 0.f, gfy = 0.f;
    if (i < n) {
        const float* oi = o + (size_t)i * 3;
        const float* di = d + (size_t)i * 3;
        const float* gr = g + (size_t)i * cols;
        float go[3], gd[3];
        if (f2) {
            ndc_backward(H, W, f2[0], f2[1], ndc_near, oi, di, gr, gr + 3, go, gd, &gfx, &gfy);
        } else {
            go[0] = gr[0]; go[1] = gr[1]; go[2] = gr[2]; gd[0] = gr[3]; gd[1] = gr[4]; gd[2] = gr[5];
        }
        if (cols > 8) {             // v = d / |d|:  g_d += (g_v - v (v . g_v)) / |d|
            const float nrm = sqrtf(di[0] * di[0] + di[1] * di[1] + di[2] * di[2]);
            const float v0 = di[0] / nrm, v1 = di[1] / nrm, v2 = di[2] / nrm;
            const float dotv = v0 * gr[8] + v1 * gr[9] + v2 * gr[10];
            gd[0] += (gr[8] - v0 * dotv) / nrm;
            gd[1] += (gr[9] - v1 * dotv) / nrm;
            gd[2] += (gr[10] - v2 * dotv) / nrm;
        }
        g_o[(size_t)i * 3] = go[0]; g_o[(size_t)i * 3 + 1] = go[1]; g_o[(size_t)i * 3 + 2] = go[2];
        g_d[(size_t)i * 3] = gd[0]; g_d[(size_t)i * 3 + 1] = gd[1]; g_d[(size_t)i * 3 + 2] = gd[2];
    }
    if (g_f2) {
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { gfx += shfl_xor(gfx, o2); gfy += shfl_xor(gfy, o2); }
        if (lane_id() == 0) { atomic_add(g_f2, gfx); atomic_add(g_f2 + 1, gfy); }
    }
}

// ---- full-image upsampling of a noise grid (CameraModel.get_ray_{o,d}_noise) -------------------
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ grid, float scale, int gh, int gw,
                                                           int H, int W, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const Taps t = bilinear_taps(i / W, i % W, gh, gw, H, W);
    const Vec3 v = scale * sample_grid(grid, gw, t);      // (interpolate) * scale, camera_model.py:24-46
    out[(size_t)i * 3] = v.x; out[(size_t)i * 3 + 1] = v.y; out[(size_t)i * 3 + 2] = v.z;
}

__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ g_out, float scale, int gh, int gw,
                                                           int H, int W, float* d_grid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const Taps t = bilinear_taps(i / W, i % W, gh, gw, H, W);
    scatter_grid(d_grid, gw, t, scale * v3(g_out[(size_t)i * 3], g_out[(size_t)i * 3 + 1], g_out[(size_t)i * 3 + 2]));
}

CamArgs make_args(const float* kps, const long long* cam_idx, int single_idx, const float* extrinsic, int n_ext,
                  const float* intr_init, const float* intr_noise, float intr_scale, int multiplicative,
                  const float* extr_init, const float* extr_noise, float extr_scale, int n_cams,
                  const float* grid_o, float scale_o, const float* grid_d, float scale_d, int gh, int gw, int H,
                  int W, int n) {
    CamArgs a;
    a.kps = kps; a.cam_idx = cam_idx; a.single_idx = single_idx; a.extrinsic = extrinsic; a.n_ext = n_ext;
    a.intr_init = intr_init; a.intr_noise = intr_noise; a.intr_scale = intr_scale; a.multiplicative = multiplicative;
    a.extr_init = extr_init; a.extr_noise = extr_noise; a.extr_scale = extr_scale; a.n_cams = n_cams;
    a.grid_o = grid_o; a.scale_o = scale_o; a.grid_d = grid_d; a.scale_d = scale_d; a.gh = gh; a.gw = gw;
    a.H = H; a.W = W; a.n = n;
    a.select = nullptr; a.dist = nullptr;
    return a;
}

int check_args(const CamArgs& a) {
    SCN_RETURN_IF(a.n < 0 || a.H < 1 || a.W < 1 || !a.intr_init || !a.intr_noise, SCN_EINVAL);
    SCN_RETURN_IF(!a.kps && !a.select && a.n != a.H * a.W, SCN_EINVAL);
    if (a.extrinsic) { SCN_RETURN_IF(a.n_ext != 1 && a.n_ext != a.n, SCN_EINVAL); }
    else {
        SCN_RETURN_IF(!a.extr_init || !a.extr_noise || a.n_cams < 1, SCN_EINVAL);
        SCN_RETURN_IF(!a.cam_idx && (a.single_idx < 0 || a.single_idx >= a.n_cams), SCN_EINVAL);
    }
    SCN_RETURN_IF((a.grid_o || a.grid_d) && (a.gh < 1 || a.gw < 1), SCN_EINVAL);
    return 0;
}

}  // namespace

extern "C" int scnerf_camera_rays_fwd(const float* kps, const long long* cam_idx, int single_idx,
                                      const float* extrinsic, int n_ext, const float* intr_init,
                                      const float* intr_noise, float intr_scale, int multiplicative,
                                      const float* extr_init, const float* extr_noise, float extr_scale,
                                      int n_cams, const float* grid_o, float scale_o, const float* grid_d,
                                      float scale_d, int gh, int gw, int H, int W, float* rays_o, float* rays_d,
                                      int n, void* stream) {
    const CamArgs a = make_args(kps, cam_idx, single_idx, extrinsic, n_ext, intr_init, intr_noise, intr_scale,
                                multiplicative, extr_init, extr_noise, extr_scale, n_cams, grid_o, scale_o, grid_d,
                                scale_d, gh, gw, H, W, n);
    const int rc = check_args(a);
    SCN_RETURN_IF(rc != 0, rc);
    SCN_RETURN_IF(!rays_o || !rays_d, SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(camera_rays_fwd_kernel, dim3(scn_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       rays_o, rays_d);
    return scn_launch_status();
}

// ---- the camera model's matrices: get_intrinsic() / get_extrinsic() (model/camera_model.py:160-192) ----------------
// K [4,4] from the four learnable intrinsics, E [C,4,4] from the 6-D rotation + translation of every camera: what the
// projected-ray-distance term and the NDC warp read each step.  As ~35 tensor ops (and ~70 in their backward) they were
// most of the 214 launches of one PRD term; here one launch each way, thread per camera (+ one for K), the same
// gram_schmidt the ray generator uses.
__global__ __launch_bounds__(64) void camera_matrices_fwd_kernel(const float* __restrict__ intr_init, const float* __restrict__ intr_noise,
                                                                 float intr_scale, int multiplicative,
                                                                 const float* __restrict__ extr_init, const float* __restrict__ extr_noise,
                                                                 float extr_scale, int n_cams, float* __restrict__ K, float* __restrict__ E) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == n_cams) {
        float p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float init = intr_init[i];
            p[i] = multiplicative ? init + intr_noise[i] * intr_scale * init : init + intr_noise[i] * intr_scale;
        }
        // intrinsic_param_to_K (camera_utils.py:191-195): identity with (0,0) = fx, (1,1) = fy, (0,2) = cx, (1,2) = cy
#pragma unroll
        for (int i = 0; i < 16; ++i) K[i] = (i % 5 == 0) ? 1.f : 0.f;
        K[0] = p[0]; K[5] = p[1]; K[2] = p[2]; K[6] = p[3];
        return;
    }
    if (c > n_cams) return;
    float p9[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) p9[k] = extr_init[(size_t)c * 9 + k] + extr_scale * extr_noise[(size_t)c * 9 + k];
    GramSchmidt s;
    const Rot R = gram_schmidt(p9, &s);
    float* e = E + (size_t)c * 16;
    e[0] = R.x.x; e[1] = R.y.x; e[2] = R.z.x; e[3] = p9[6];
    e[4] = R.x.y; e[5] = R.y.y; e[6] = R.z.y; e[7] = p9[7];
    e[8] = R.x.z; e[9] = R.y.z; e[10] = R.z.z; e[11] = p9[8];
    e[12] = 0.f; e[13] = 0.f; e[14] = 0.f; e[15] = 1.f;
}

__global__ __launch_bounds__(64) void camera_matrices_bwd_kernel(const float* __restrict__ intr_init, float intr_scale, int multiplicative,
                                                                 const float* __restrict__ extr_init, const float* __restrict__ extr_noise,
                                                                 float extr_scale, int n_cams, const float* __restrict__ gK,
                                                                 const float* __restrict__ gE, float* __restrict__ d_intr_noise,
                                                                 float* __restrict__ d_extr_noise) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == n_cams) {
        if (d_intr_noise) {
            const int at[4] = {0, 5, 2, 6};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                d_intr_noise[i] = gK ? gK[at[i]] * intr_scale * (multiplicative ? intr_init[i] : 1.f) : 0.f;
        }
        return;
    }
    if (c > n_cams || !d_extr_noise) return;
    float* d = d_extr_noise + (size_t)c * 9;
    if (!gE) {
#pragma unroll
        for (int k = 0; k < 9; ++k) d[k] = 0.f;
        return;
    }
    float p9[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) p9[k] = extr_init[(size_t)c * 9 + k] + extr_scale * extr_noise[(size_t)c * 9 + k];
    GramSchmidt s;
    gram_schmidt(p9, &s);
    const float* g = gE + (size_t)c * 16;
    float g6[6];
    gram_schmidt_bwd(s, v3(g[0], g[4], g[8]), v3(g[1], g[5], g[9]), v3(g[2], g[6], g[10]), g6);
#pragma unroll
    for (int k = 0; k < 6; ++k) d[k] = extr_scale * g6[k];
    d[6] = extr_scale * g[3]; d[7] = extr_scale * g[7]; d[8] = extr_scale * g[11];
}

extern "C" int scnerf_camera_matrices_fwd(const float* intr_init, const float* intr_noise, float intr_scale,
                                          int multiplicative, const float* extr_init, const float* extr_noise,
                                          float extr_scale, int n_cams, float* K, float* E, void* stream) {
    SCN_RETURN_IF(!intr_init || !intr_noise || !extr_init || !extr_noise || !K || !E || n_cams < 0, SCN_EINVAL);
    hipLaunchKernelGGL(camera_matrices_fwd_kernel, dim3(scn_ceil_div(n_cams + 1, 64)), dim3(64), 0, (hipStream_t)stream,
                       intr_init, intr_noise, intr_scale, multiplicative, extr_init, extr_noise, extr_scale, n_cams, K, E);
    return scn_launch_status();
}

extern "C" int scnerf_camera_matrices_bwd(const float* intr_init, float intr_scale, int multiplicative,
                                          const float* extr_init, const float* extr_noise, float extr_scale, int n_cams,
                                          const float* g_K, const float* g_E, float* d_intr_noise, float* d_extr_noise,
                                          void* stream) {
    SCN_RETURN_IF(!intr_init || !extr_init || !extr_noise || n_cams < 0 || (!d_intr_noise && !d_extr_noise), SCN_EINVAL);
    hipLaunchKernelGGL(camera_matrices_bwd_kernel, dim3(scn_ceil_div(n_cams + 1, 64)), dim3(64), 0, (hipStream_t)stream,
                       intr_init, intr_scale, multiplicative, extr_init, extr_noise, extr_scale, n_cams, g_K, g_E,
                       d_intr_noise, d_extr_noise);
    return scn_launch_status();
}

extern "C" long long scnerf_camera_bwd_workspace_floats(int n_slots) { return 4 + 12LL * (n_slots < 1 ? 1 : n_slots); }

extern "C" int scnerf_camera_rays_bwd(const float* kps, const long long* cam_idx, int single_idx,
                                      const float* extrinsic, int n_ext, const float* intr_init,
                                      const float* intr_noise, float intr_scale, int multiplicative,
                                      const float* extr_init, const float* extr_noise, float extr_scale,
                                      int n_cams, const float* grid_o, float scale_o, const float* grid_d,
                                      float scale_d, int gh, int gw, int H, int W, const float* g_o,
                                      const float* g_d, float* d_intr_noise, float* d_extr_noise, float* d_grid_o,
                                      float* d_grid_d, float* d_extrinsic, float* workspace, int n, void* stream) {
    const CamArgs a = make_args(kps, cam_idx, single_idx, extrinsic, n_ext, intr_init, intr_noise, intr_scale,
                                multiplicative, extr_init, extr_noise, extr_scale, n_cams, grid_o, scale_o, grid_d,
                                scale_d, gh, gw, H, W, n);
    const int rc = check_args(a);
    SCN_RETURN_IF(rc != 0, rc);
    SCN_RETURN_IF(!workspace, SCN_EINVAL);
    hipStream_t st = (hipStream_t)stream;
    const int slots = extrinsic ? n_ext : n_cams;
    SCN_HIP(hipMemsetAsync(workspace, 0, sizeof(float) * (4 + 12 * (size_t)slots), st));
    if (d_grid_o) SCN_HIP(hipMemsetAsync(d_grid_o, 0, sizeof(float) * 3 * (size_t)gh * gw, st));
    if (d_grid_d && d_grid_d != d_grid_o) SCN_HIP(hipMemsetAsync(d_grid_d, 0, sizeof(float) * 3 * (size_t)gh * gw, st));
    if (d_extr_noise) SCN_HIP(hipMemsetAsync(d_extr_noise, 0, sizeof(float) * 9 * (size_t)n_cams, st));
    if (n > 0) {
        // per-ray poses (n_ext == n) have one slot per ray: nothing to pre-reduce
        const int lds_slots = (extrinsic && n_ext > 1) || slots > 1024 ? 0 : slots;
        hipLaunchKernelGGL(camera_rays_bwd_kernel, dim3(scn_ceil_div(n, 256)), dim3(256), (size_t)lds_slots * 12 * sizeof(float),
                           st, a, g_o, g_d, workspace, d_grid_o, d_grid_d, (float*)nullptr, lds_slots);
    }
    hipLaunchKernelGGL(camera_finish_kernel, dim3(scn_ceil_div(slots, 64)), dim3(64), 0, st, a, workspace,
                       d_intr_noise, d_extr_noise, d_extrinsic);
    return scn_launch_status();
}

// ---- NeRF++ generator: same camera model, pixel centres of selected pixels, optional radial distortion ----
extern "C" int scnerf_npp_camera_rays_fwd(const long long* select, const float* dist2, int camera_idx,
                                          const float* extrinsic, const float* intr_init, const float* intr_noise,
                                          float intr_scale, int multiplicative, const float* extr_init,
                                          const float* extr_noise, float extr_scale, int n_cams,
                                          const float* grid_o, float scale_o, const float* grid_d, float scale_d,
                                          int gh, int gw, int H, int W, float* rays_o, float* rays_d, int n,
                                          void* stream) {
    CamArgs a = make_args(nullptr, nullptr, camera_idx, extrinsic, extrinsic ? 1 : 0, intr_init, intr_noise,
                          intr_scale, multiplicative, extr_init, extr_noise, extr_scale, n_cams, grid_o, scale_o,
                          grid_d, scale_d, gh, gw, H, W, n);
    a.select = select; a.dist = dist2;
    SCN_RETURN_IF(!select || !rays_o || !rays_d, SCN_EINVAL);
    const int rc = check_args(a);
    SCN_RETURN_IF(rc != 0, rc);
    if (n == 0) return 0;
    hipLaunchKernelGGL(camera_rays_fwd_kernel, dim3(scn_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       rays_o, rays_d);
    return scn_launch_status();
}

extern "C" int scnerf_npp_camera_rays_bwd(const long long* select, const float* dist2, int camera_idx,
                                          const float* extrinsic, const float* intr_init, const float* intr_noise,
                                          float intr_scale, int multiplicative, const float* extr_init,
                                          const float* extr_noise, float extr_scale, int n_cams,
                                          const float* grid_o, float scale_o, const float* grid_d, float scale_d,
                                          int gh, int gw, int H, int W, const float* g_o, const float* g_d,
                                          float* d_intr_noise, float* d_extr_noise, float* d_grid_o,
                                          float* d_grid_d, float* d_extrinsic, float* d_dist2, float* workspace,
                                          int n, void* stream) {
    CamArgs a = make_args(nullptr, nullptr, camera_idx, extrinsic, extrinsic ? 1 : 0, intr_init, intr_noise,
                          intr_scale, multiplicative, extr_init, extr_noise, extr_scale, n_cams, grid_o, scale_o,
                          grid_d, scale_d, gh, gw, H, W, n);
    a.select = select; a.dist = dist2;
    SCN_RETURN_IF(!select || !workspace || (dist2 && !d_dist2), SCN_EINVAL);
    const int rc = check_args(a);
    SCN_RETURN_IF(rc != 0, rc);
    hipStream_t st = (hipStream_t)stream;
    const int slots = extrinsic ? 1 : n_cams;
    SCN_HIP(hipMemsetAsync(workspace, 0, sizeof(float) * (4 + 12 * (size_t)slots), st));
    if (d_dist2) SCN_HIP(hipMemsetAsync(d_dist2, 0, 2 * sizeof(float), st));      // accumulated into directly
    if (d_grid_o) SCN_HIP(hipMemsetAsync(d_grid_o, 0, sizeof(float) * 3 * (size_t)gh * gw, st));
    if (d_grid_d && d_grid_d != d_grid_o) SCN_HIP(hipMemsetAsync(d_grid_d, 0, sizeof(float) * 3 * (size_t)gh * gw, st));
    if (d_extr_noise) SCN_HIP(hipMemsetAsync(d_extr_noise, 0, sizeof(float) * 9 * (size_t)n_cams, st));
    if (n > 0) {
        const int lds_slots = slots > 1024 ? 0 : slots;
        hipLaunchKernelGGL(camera_rays_bwd_kernel, dim3(scn_ceil_div(n, 256)), dim3(256), (size_t)lds_slots * 12 * sizeof(float),
                           st, a, g_o, g_d, workspace, d_grid_o, d_grid_d, dist2 ? d_dist2 : (float*)nullptr, lds_slots);
    }
    hipLaunchKernelGGL(camera_finish_kernel, dim3(scn_ceil_div(slots, 64)), dim3(64), 0, st, a, workspace,
                       d_intr_noise, d_extr_noise, d_extrinsic);
    return scn_launch_status();
}

extern "C" int scnerf_pinhole_rays(const float* kps, int kps_stride, const float* c2w, float focal, int H, int W,
                                   float* rays_o, float* rays_d, int n, void* stream) {
    SCN_RETURN_IF(!c2w || !rays_o || !rays_d || n < 0 || (kps && kps_stride < 2) || (!kps && n != H * W), SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(pinhole_rays_kernel, dim3(scn_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, kps,
                       kps_stride, c2w, focal, H, W, rays_o, rays_d, n);
    return scn_launch_status();
}

extern "C" int scnerf_ndc_fwd(int H, int W, const float* focal_xy, float near, const float* rays_o,
                              const float* rays_d, float* ndc_o, float* ndc_d, int n, void* stream) {
    SCN_RETURN_IF(!focal_xy || !rays_o || !rays_d || !ndc_o || !ndc_d || n < 0, SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(ndc_fwd_kernel, dim3(scn_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, H, W,
                       focal_xy, near, rays_o, rays_d, ndc_o, ndc_d, n);
    return scn_launch_status();
}

extern "C" int scnerf_ndc_bwd(int H, int W, const float* focal_xy, float near, const float* rays_o,
                              const float* rays_d, const float* g_ndc_o, const float* g_ndc_d, float* g_rays_o,
                              float* g_rays_d, float* g_focal_xy, int n, void* stream) {
    SCN_RETURN_IF(!focal_xy || !rays_o || !rays_d || !g_rays_o || !g_rays_d || n < 0, SCN_EINVAL);
    hipStream_t st = (hipStream_t)stream;
    if (g_focal_xy) SCN_HIP(hipMemsetAsync(g_focal_xy, 0, 2 * sizeof(float), st));
    if (n == 0) return 0;
    hipLaunchKernelGGL(ndc_bwd_kernel, dim3(scn_ceil_div(n, 256)), dim3(256), 0, st, H, W, focal_xy, near, rays_o,
                       rays_d, g_ndc_o, g_ndc_d, g_rays_o, g_rays_d, g_focal_xy, n);
    return scn_launch_status();
}

extern "C" int scnerf_pack_rays_fwd(int H, int W, const float* focal_xy, float ndc_near, const float* rays_o,
                                    const float* rays_d, float near, float far, int cols, float* ray_batch, int n,
                                    void* stream) {
    SCN_RETURN_IF(!rays_o || !rays_d || !ray_batch || n < 0 || (cols != 8 && cols != 11), SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_rays_fwd_kernel, dim3(scn_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, H, W, focal_xy,
                       ndc_near, rays_o, rays_d, near, far, cols, ray_batch, n);
    return scn_launch_status();
}

extern "C" int scnerf_pack_rays_bwd(int H, int W, const float* focal_xy, float ndc_near, const float* rays_o,
                                    const float* rays_d, int cols, const float* g_ray_batch, float* g_rays_o,
                                    float* g_rays_d, float* g_focal_xy, int n, void* stream) {
    SCN_RETURN_IF(!rays_o || !rays_d || !g_ray_batch || !g_rays_o || !g_rays_d || n < 0 || (cols != 8 && cols != 11), SCN_EINVAL);
    SCN_RETURN_IF(g_focal_xy && !focal_xy, SCN_EINVAL);
    hipStream_t st = (hipStream_t)stream;
    if (g_focal_xy) SCN_HIP(hipMemsetAsync(g_focal_xy, 0, 2 * sizeof(float), st));
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_rays_bwd_kernel, dim3(scn_ceil_div(n, 256)), dim3(256), 0, st, H, W, focal_xy, ndc_near, rays_o,
                       rays_d, cols, g_ray_batch, g_rays_o, g_rays_d, g_focal_xy, n);
    return scn_launch_status();
}

extern "C" int scnerf_upsample_grid_fwd(const float* grid, float scale, int gh, int gw, int H, int W, float* out,
                                        void* stream) {
    SCN_RETURN_IF(!grid || !out || gh < 1 || gw < 1 || H < 1 || W < 1, SCN_EINVAL);
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(scn_ceil_div((long long)H * W, 256)), dim3(256), 0,
                       (hipStream_t)stream, grid, scale, gh, gw, H, W, out);
    return scn_launch_status();
}

extern "C" int scnerf_upsample_grid_bwd(const float* g_out, float scale, int gh, int gw, int H, int W,
                                        float* d_grid, void* stream) {
    SCN_RETURN_IF(!g_out || !d_grid || gh < 1 || gw < 1 || H < 1 || W < 1, SCN_EINVAL);
    hipStream_t st = (hipStream_t)stream;
    SCN_HIP(hipMemsetAsync(d_grid, 0, sizeof(float) * 3 * (size_t)gh * gw, st));
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(scn_ceil_div((long long)H * W, 256)), dim3(256), 0, st, g_out,
                       scale, gh, gw, H, W, d_grid);
    return scn_launch_status();
}
