// prd_loss.hip -- projected-ray-distance (PRD) loss of SCNeRF, forward + backward, one thread per
// matched key-point pair.
//
// Replaces proj_ray_dist_loss_single (/root/reference model/ray_dist_loss.py:22-246) and what autograd
// derives from its ~60 tensor ops: closest points of two rays, re-projection of each into the other
// view through E^-1 and K (K[0][0] negated for NeRF's axes, :102-105), squared pixel error, chirality
// (t > 0) and threshold / finiteness masks, two-way masked mean.  <= ~1k matches per call: latency, not
// bandwidth -- the point is one launch each way instead of dozens of tiny kernels.
#include <scn_wave.h>

#include "launch.h"
#include "scnerf_hip.h"

namespace {

using namespace scn;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 ld3(const float* p, long i) { return mk(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]); }

struct Cam { float R[3][3]; V3 t; };     // R[a][b] = E[a][b], t = E[:3, 3]
__device__ __forceinline__ Cam load_cam(const float* E) {
    Cam c;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) c.R[a][b] = E[a * 4 + b];
    c.t = mk(E[3], E[7], E[11]);
    return c;
}
// q = R^T (p - t)   (the 4th row of E^-1 gives 1)
__device__ __forceinline__ V3 to_cam(const Cam& c, V3 p) {
    const V3 d = p - c.t;
    return mk(c.R[0][0] * d.x + c.R[1][0] * d.y + c.R[2][0] * d.z,
              c.R[0][1] * d.x + c.R[1][1] * d.y + c.R[2][1] * d.z,
              c.R[0][2] * d.x + c.R[1][2] * d.y + c.R[2][2] * d.z);
}

struct PrdArgs {
    const float* kps0; const float* kps1;            // [m,2]
    const float* o0; const float* d0; const float* o1; const float* d1;   // [m,3]
    const float* K;                                   // [4,4]
    const float* E;                                   // [2,4,4]
    float eps, threshold;
    int negate_fx, eval_mode, m;
};

struct Match {
    V3 o0, o1, rd0, rd1, d0, d1, w, p0, p1, q01, q10;
    float n0, n1, s0, s1, a, b, r, den, t0, t1;
    float n01[3], n10[3], u01[2], u10[2], L0, L1;
    bool chir;
};

__device__ __forceinline__ void kmat(const PrdArgs& a, float Kk[3][4]) {
    for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 4; ++k) Kk[j][k] = a.K[j * 4 + k];
    if (a.negate_fx) Kk[0][0] = -Kk[0][0];
}

__device__ __forceinline__ void match_forward(const PrdArgs& a, const float Kk[3][4], const Cam& c0, const Cam& c1,
                                              int i, Match* f) {
    f->o0 = ld3(a.o0, i); f->o1 = ld3(a.o1, i); f->rd0 = ld3(a.d0, i); f->rd1 = ld3(a.d1, i);
    f->n0 = sqrtf(dot(f->rd0, f->rd0)); f->n1 = sqrtf(dot(f->rd1, f->rd1));
    f->s0 = f->n0 + a.eps; f->s1 = f->n1 + a.eps;
    f->d0 = mk(f->rd0.x / f->s0, f->rd0.y / f->s0, f->rd0.z / f->s0);
    f->d1 = mk(f->rd1.x / f->s1, f->rd1.y / f->s1, f->rd1.z / f->s1);
    f->w = f->o0 - f->o1;
    f->a = dot(f->d0, f->w); f->b = dot(f->d1, f->w); f->r = dot(f->d0, f->d1);
    f->den = f->r * f->r - 1.f + a.eps;
    f->t0 = (f->a - f->r * f->b) / f->den;
    f->t1 = (-f->b + f->r * f->a) / f->den;          // = (d1.(o1-o0) - r d0.(o1-o0)) / den
    f->p0 = f->t0 * f->d0 + f->o0;
    f->p1 = f->t1 * f->d1 + f->o1;
    f->q01 = to_cam(c1, f->p0);                      // point on ray 0 seen from camera 1
    f->q10 = to_cam(c0, f->p1);
    const float q01[3] = {f->q01.x, f->q01.y, f->q01.z}, q10[3] = {f->q10.x, f->q10.y, f->q10.z};
    for (int j = 0; j < 3; ++j) {
        f->n01[j] = Kk[j][0] * q01[0] + Kk[j][1] * q01[1] + Kk[j][2] * q01[2] + Kk[j][3];
        f->n10[j] = Kk[j][0] * q10[0] + Kk[j][1] * q10[1] + Kk[j][2] * q10[2] + Kk[j][3];
    }
    for (int j = 0; j < 2; ++j) {
        f->u01[j] = f->n01[j] / (f->n01[2] + a.eps);
        f->u10[j] = f->n10[j] / (f->n10[2] + a.eps);
    }
    f->chir = f->t0 > 0.f && f->t1 > 0.f;
    const float e0x = f->u10[0] - a.kps0[i * 2], e0y = f->u10[1] - a.kps0[i * 2 + 1];
    const float e1x = f->u01[0] - a.kps1[i * 2], e1y = f->u01[1] - a.kps1[i * 2 + 1];
    f->L0 = e0x * e0x + e0y * e0y;
    f->L1 = e1x * e1x + e1y * e1y;
}

__device__ __forceinline__ bool finite_f(float x) { return fabsf(x) <= 3.402823466e38f; }   // false for nan / inf

// sums[0..5] = sum L0 (masked), count0, sum L1, count1, count(both), count(chirality)
__global__ __launch_bounds__(256) void prd_fwd_kernel(PrdArgs a, float* sums) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < a.m) {
        float Kk[3][4];
        kmat(a, Kk);
        const Cam c0 = load_cam(a.E), c1 = load_cam(a.E + 16);
        Match f;
        match_forward(a, Kk, c0, c1, i, &f);
        if (f.chir) {
            const bool ok0 = f.L0 < a.threshold && finite_f(f.L0), ok1 = f.L1 < a.threshold && finite_f(f.L1);
            if (!a.eval_mode) {
                if (ok0) { s[0] = f.L0; s[1] = 1.f; }
                if (ok1) { s[2] = f.L1; s[3] = 1.f; }
                if (ok0 && ok1) s[4] = 1.f;
            } else {        // val / test: invalid entries count as the threshold, mean over the chirality-valid
                const bool bad0 = f.L0 > a.threshold || !finite_f(f.L0), bad1 = f.L1 > a.threshold || !finite_f(f.L1);
                s[0] = bad0 ? a.threshold : f.L0; s[1] = 1.f;
                s[2] = bad1 ? a.threshold : f.L1; s[3] = 1.f;
            }
            s[5] = 1.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float v = s[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += shfl_xor(v, o);
        if (lane_id() == 0 && v != 0.f) atomic_add(sums + k, v);
    }
}

// keep[i] = both re-projection errors below `threshold` and both closest points in front of their cameras
// (filter_matches_with_gt, model/prd_evaluation.py:189-332, which fixes threshold = 1 pixel^2)
__global__ __launch_bounds__(256) void prd_filter_kernel(PrdArgs a, unsigned char* keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.m) return;
    float Kk[3][4];
    kmat(a, Kk);
    const Cam c0 = load_cam(a.E), c1 = load_cam(a.E + 16);
    Match f;
    match_forward(a, Kk, c0, c1, i, &f);
    keep[i] = (f.chir && f.L0 < a.threshold && f.L1 < a.threshold) ? 1 : 0;
}

__global__ void prd_finish_kernel(const float* sums, float* loss, float* n_match) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *loss = 0.5f * (sums[0] / sums[1] + sums[2] / sums[3]);     // empty selection -> nan, as the reference
        if (n_match) *n_match = sums[4];
    }
}

// acc: [0..11] dK (3x4, w.r.t. the matrix as passed, sign of K[0][0] already undone), [12..35] dE (2 x 12: R row-major 9 + t 3)
__global__ __launch_bounds__(256) void prd_bwd_kernel(PrdArgs a, const float* sums, const float* g_loss,
                                                      float* g_o0, float* g_d0, float* g_o1, float* g_d1, float* acc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float gk[12], ge[24];
#pragma unroll
    for (int k = 0; k < 12; ++k) gk[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 24; ++k) ge[k] = 0.f;
    if (i < a.m) {
        float Kk[3][4];
        kmat(a, Kk);
        const Cam c0 = load_cam(a.E), c1 = load_cam(a.E + 16);
        Match f;
        match_forward(a, Kk, c0, c1, i, &f);
        float gL0 = 0.f, gL1 = 0.f;
        if (f.chir && !a.eval_mode) {
            const float g = 0.5f * g_loss[0];
            if (f.L0 < a.threshold && finite_f(f.L0)) gL0 = g / sums[1];
            if (f.L1 < a.threshold && finite_f(f.L1)) gL1 = g / sums[3];
        }
        V3 go0 = mk(0, 0, 0), go1 = mk(0, 0, 0), gd0 = mk(0, 0, 0), gd1 = mk(0, 0, 0);
        float g_t0 = 0.f, g_t1 = 0.f;
        // two re-projections: (u01: p0 through camera 1, error vs kps1) and (u10: p1 through camera 0, vs kps0)
        for (int side = 0; side < 2; ++side) {
            const float gL = side == 0 ? gL1 : gL0;
            if (gL == 0.f) continue;
            const float* u = side == 0 ? f.u01 : f.u10;
            const float* n = side == 0 ? f.n01 : f.n10;
            const float* kp = side == 0 ? a.kps1 + (size_t)i * 2 : a.kps0 + (size_t)i * 2;
            const V3 q = side == 0 ? f.q01 : f.q10;
            const Cam& c = side == 0 ? c1 : c0;
            const V3 p = side == 0 ? f.p0 : f.p1;
            const float gu0 = 2.f * (u[0] - kp[0]) * gL, gu1 = 2.f * (u[1] - kp[1]) * gL;
            const float dn = n[2] + a.eps;
            const float gn[3] = {gu0 / dn, gu1 / dn, -(gu0 * u[0] + gu1 * u[1]) / dn};
            const float qq[3] = {q.x, q.y, q.z};
            float gq[3] = {0.f, 0.f, 0.f};
            for (int j = 0; j < 3; ++j) {
                for (int k = 0; k < 3; ++k) {
                    gk[j * 4 + k] += gn[j] * qq[k];
                    gq[k] += gn[j] * Kk[j][k];
                }
                gk[j * 4 + 3] += gn[j];
            }
            // q = R^T (p - t):  g_p = R g_q,  g_t = -R g_q,  g_R[a][b] = (p - t)_a g_q[b]
            const V3 dp = p - c.t;
            const float dpa[3] = {dp.x, dp.y, dp.z};
            float gp[3];
            for (int aa = 0; aa < 3; ++aa) {
                gp[aa] = c.R[aa][0] * gq[0] + c.R[aa][1] * gq[1] + c.R[aa][2] * gq[2];
                for (int bb = 0; bb < 3; ++bb) ge[(side == 0 ? 12 : 0) + aa * 3 + bb] += dpa[aa] * gq[bb];
                ge[(side == 0 ? 12 : 0) + 9 + aa] -= gp[aa];
            }
            const V3 gpv = mk(gp[0], gp[1], gp[2]);
            if (side == 0) { g_t0 += dot(gpv, f.d0); gd0 = gd0 + f.t0 * gpv; go0 = go0 + gpv; }
            else { g_t1 += dot(gpv, f.d1); gd1 = gd1 + f.t1 * gpv; go1 = go1 + gpv; }
        }
        // t0 = (a - r b)/den, t1 = (-b + r a)/den, den = r^2 - 1 + eps
        float g_a = g_t0 / f.den + f.r * g_t1 / f.den;
        float g_b = -f.r * g_t0 / f.den - g_t1 / f.den;
        float g_r = -f.b * g_t0 / f.den + f.a * g_t1 / f.den;
        const float g_den = -(f.t0 * g_t0 + f.t1 * g_t1) / f.den;
        g_r += 2.f * f.r * g_den;
        V3 gw = g_a * f.d0 + g_b * f.d1;
        gd0 = gd0 + g_a * f.w + g_r * f.d1;
        gd1 = gd1 + g_b * f.w + g_r * f.d0;
        go0 = go0 + gw;
        go1 = go1 - gw;
        // d = rd / (|rd| + eps)
        V3 grd0 = (1.f / f.s0) * gd0, grd1 = (1.f / f.s1) * gd1;
        if (f.n0 > 0.f) grd0 = grd0 - (dot(gd0, f.rd0) / (f.s0 * f.s0 * f.n0)) * f.rd0;
        if (f.n1 > 0.f) grd1 = grd1 - (dot(gd1, f.rd1) / (f.s1 * f.s1 * f.n1)) * f.rd1;
        g_o0[i * 3] = go0.x; g_o0[i * 3 + 1] = go0.y; g_o0[i * 3 + 2] = go0.z;
        g_o1[i * 3] = go1.x; g_o1[i * 3 + 1] = go1.y; g_o1[i * 3 + 2] = go1.z;
        g_d0[i * 3] = grd0.x; g_d0[i * 3 + 1] = grd0.y; g_d0[i * 3 + 2] = grd0.z;
        g_d1[i * 3] = grd1.x; g_d1[i * 3 + 1] = grd1.y; g_d1[i * 3 + 2] = grd1.z;
    }
    if (a.negate_fx) gk[0] = -gk[0];
#pragma unroll
    for (int k = 0; k < 36; ++k) {
        float v = k < 12 ? gk[k] : ge[k - 12];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += shfl_xor(v, o);
        if (lane_id() == 0 && v != 0.f) atomic_add(acc + k, v);
    }
}

// accumulators -> dK [4,4] and dE [2,4,4]
__global__ void prd_unpack_kernel(const float* acc, float* dK, float* dE) {
    const int t = threadIdx.x;
    if (t < 16) dK[t] = (t < 12) ? acc[t] : 0.f;
    if (t < 32) {
        const int c = t / 16, e = t % 16, row = e / 4, col = e % 4;
        float v = 0.f;
        if (row < 3) v = col < 3 ? acc[12 + c * 12 + row * 3 + col] : acc[12 + c * 12 + 9 + row];
        dE[t] = v;
    }
}

PrdArgs make(const float* kps0, const float* kps1, const float* o0, const float* d0, const float* o1, const float* d1,
             const float* K, const float* E, float eps, float thr, int neg, int eval_mode, int m) {
    PrdArgs a;
    a.kps0 = kps0; a.kps1 = kps1; a.o0 = o0; a.d0 = d0; a.o1 = o1; a.d1 = d1; a.K = K; a.E = E;
    a.eps = eps; a.threshold = thr; a.negate_fx = neg; a.eval_mode = eval_mode; a.m = m;
    return a;
}

}  // namespace

extern "C" int scnerf_prd_loss_fwd(const float* kps0, const float* kps1, const float* rays0_o, const float* rays0_d,
                                   const float* rays1_o, const float* rays1_d, const float* K, const float* E2,
                                   float eps, float threshold, int negate_fx, int eval_mode, int m, float* sums6,
                                   float* loss, float* n_match, void* stream) {
    SCN_RETURN_IF(!kps0 || !kps1 || !rays0_o || !rays0_d || !rays1_o || !rays1_d || !K || !E2 || !sums6 || !loss || m < 0, SCN_EINVAL);
    hipStream_t st = (hipStream_t)stream;
    SCN_HIP(hipMemsetAsync(sums6, 0, 6 * sizeof(float), st));
    const PrdArgs a = make(kps0, kps1, rays0_o, rays0_d, rays1_o, rays1_d, K, E2, eps, threshold, negate_fx, eval_mode, m);
    if (m > 0) hipLaunchKernelGGL(prd_fwd_kernel, dim3(scn_ceil_div(m, 256)), dim3(256), 0, st, a, sums6);
    hipLaunchKernelGGL(prd_finish_kernel, dim3(1), dim3(64), 0, st, sums6, loss, n_match);
    return scn_launch_status();
}

extern "C" int scnerf_prd_loss_bwd(const float* kps0, const float* kps1, const float* rays0_o, const float* rays0_d,
                                   const float* rays1_o, const float* rays1_d, const float* K, const float* E2,
                                   float eps, float threshold, int negate_fx, int m, const float* sums6,
                                   const float* g_loss, float* g_rays0_o, float* g_rays0_d, float* g_rays1_o,
                                   float* g_rays1_d, float* g_K, float* g_E2, float* workspace36, void* stream) {
    SCN_RETURN_IF(!kps0 || !kps1 || !rays0_o || !rays0_d || !rays1_o || !rays1_d || !K || !E2 || !sums6 || !g_loss, SCN_EINVAL);
    SCN_RETURN_IF(!g_rays0_o || !g_rays0_d || !g_rays1_o || !g_rays1_d || !g_K || !g_E2 || !workspace36 || m < 0, SCN_EINVAL);
    hipStream_t st = (hipStream_t)stream;
    SCN_HIP(hipMemsetAsync(workspace36, 0, 36 * sizeof(float), st));
    const PrdArgs a = make(kps0, kps1, rays0_o, rays0_d, rays1_o, rays1_d, K, E2, eps, threshold, negate_fx, 0, m);
    if (m > 0)
        hipLaunchKernelGGL(prd_bwd_kernel, dim3(scn_ceil_div(m, 256)), dim3(256), 0, st, a, sums6, g_loss, g_rays0_o,
                           g_rays0_d, g_rays1_o, g_rays1_d, workspace36);
    hipLaunchKernelGGL(prd_unpack_kernel, dim3(1), dim3(64), 0, st, workspace36, g_K, g_E2);
    return scn_launch_status();
}

extern "C" int scnerf_prd_filter(const float* kps0, const float* kps1, const float* rays0_o, const float* rays0_d,
                                 const float* rays1_o, const float* rays1_d, const float* K, const float* E2,
                                 float eps, float threshold, int negate_fx, int m, unsigned char* keep, void* stream) {
    SCN_RETURN_IF(!kps0 || !kps1 || !rays0_o || !rays0_d || !rays1_o || !rays1_d || !K || !E2 || !keep || m < 0, SCN_EINVAL);
    const PrdArgs a = make(kps0, kps1, rays0_o, rays0_d, rays1_o, rays1_d, K, E2, eps, threshold, negate_fx, 1, m);
    if (m > 0) hipLaunchKernelGGL(prd_filter_kernel, dim3(scn_ceil_div(m, 256)), dim3(256), 0, (hipStream_t)stream, a, keep);
    return scn_launch_status();
}
