// sampling.hip -- stratified + hierarchical (inverse-CDF) sampling for gfx950.
//
// One 64-lane wave owns one ray; the per-ray arrays (cdf, bins, u, merge buffer) live in
// LDS.  The inverse-CDF lookup is a wave-ballot upper-bound search (one v_cmp + s_bcnt1
// per sample for <= 64 knots).  Arithmetic is rounded op by op (the file is compiled with
// -ffp-contract=off) so that, fed the same weights, it reproduces the reference's fp32
// tensor arithmetic; the two reductions whose rounding order matters are restated
// explicitly: the pdf normaliser uses ATen's row-sum order and the cdf / is accumulated in
// fp64 like ATen's CPU cumsum (oracle/scnerf_oracle.py documents and pins both).
#include <scn_wave.h>

#include "aten_sum.h"
#include "launch.h"
#include "ray_stage.h"
#include "scnerf_hip.h"

namespace {

using namespace scn;

constexpr int kRaysPerBlock = 4;  // 4 waves / 256 threads

// ---- per-wave pieces (lds arrays are private to the wave; block_sync() orders them) ----

// cdf[0..nb) from weights w_in[0..nb-1) (already offset); s_w is scratch of >= nb floats
__device__ void build_cdf(const float* w_in, int nb, float* s_w, float* s_cdf, int lane) {
    const int m = nb - 1;
    for (int k = lane; k < m; k += kWave) s_w[k] = w_in[k] + 1e-5f;
    block_sync();
    if (lane == 0) {
        const float tot = aten_rowsum(s_w, m);
        double run = 0.0;
        s_cdf[0] = 0.f;
#pragma unroll 1
        for (int k = 0; k < m; ++k) {
            const float pdf = s_w[k] / tot;
            run += (double)pdf;
            s_cdf[k + 1] = (float)run;
        }
    }
    block_sync();
}

// upper bound (count of cdf entries <= u) for the ns samples of this ray; inds into s_ind
__device__ void search_right(const float* s_cdf, int nb, const float* s_u, int ns, int* s_ind,
                             int lane, bool side_left) {
    if (nb <= kWave) {
        const float c = lane < nb ? s_cdf[lane] : 0.f;
#pragma unroll 1      // (fully unrolled these loops cost 248 VGPRs + scratch: one wave per SIMD)
        for (int j = 0; j < ns; ++j) {
            const float uq = s_u[j];  // LDS broadcast
            const bool le = side_left ? (c < uq) : (c <= uq);
            const unsigned long long m = ballot(lane < nb && le);
            if (lane == (j & 63)) s_ind[j] = popcount64(m);
        }
    } else {
        for (int j = lane; j < ns; j += kWave) {
            const float uq = s_u[j];
            int lo = 0, hi = nb;  // first index with cdf[idx] > u  (>= for side_left)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const float c = s_cdf[mid];
                const bool go_right = side_left ? (c < uq) : (c <= uq);
                if (go_right) lo = mid + 1; else hi = mid;
            }
            s_ind[j] = lo;
        }
    }
}

__device__ __forceinline__ float invert_cdf(const float* s_cdf, const float* s_bins, int nb,
                                            float u, int ind) {
    const int below = max(0, ind - 1);
    const int above = min(nb - 1, ind);
    const float c0 = s_cdf[below], c1 = s_cdf[above];
    const float b0 = s_bins[below], b1 = s_bins[above];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.f;
    const float t = (u - c0) / denom;
    return b0 + t * (b1 - b0);
}

// ------------------------------------------------------------------- kernels ----------
__global__ __launch_bounds__(256) void searchsorted_kernel(
    const float* __restrict__ a, const float* __restrict__ v, int64_t* __restrict__ out, int nrow,
    int nrow_a, int nrow_v, int na, int nv, int side_left, int lds_per_wave) {
    float* lds = dynamic_lds<float>() + (size_t)wave_id() * lds_per_wave;
    float* s_a = lds;
    float* s_v = s_a + na;
    int* s_ind = reinterpret_cast<int*>(s_v + nv);
    const int lane = lane_id();
    int row = blockIdx.x * kRaysPerBlock + wave_id();
    const bool live = row < nrow;
    if (!live) row = nrow - 1;
    const float* ar = a + (size_t)(nrow_a == 1 ? 0 : row) * na;
    const float* vr = v + (size_t)(nrow_v == 1 ? 0 : row) * nv;
    for (int k = lane; k < na; k += kWave) s_a[k] = ar[k];
    for (int k = lane; k < nv; k += kWave) s_v[k] = vr[k];
    block_sync();
    search_right(s_a, na, s_v, nv, s_ind, lane, side_left != 0);
    block_sync();
    if (live)
        for (int j = lane; j < nv; j += kWave) out[(size_t)row * nv + j] = (int64_t)s_ind[j];
}

__global__ __launch_bounds__(256) void sample_pdf_kernel(
    const float* __restrict__ bins, const float* __restrict__ weights, const float* __restrict__ u,
    int u_row_stride, float* __restrict__ samples, int64_t* __restrict__ inds,
    float* __restrict__ cdf_out, int n, int nb, int ns, int lds_per_wave) {
    float* lds = dynamic_lds<float>() + (size_t)wave_id() * lds_per_wave;
    float* s_w = lds;
    float* s_cdf = s_w + nb;
    float* s_bins = s_cdf + nb;
    float* s_u = s_bins + nb;
    int* s_ind = reinterpret_cast<int*>(s_u + ns);
    const int lane = lane_id();
    int ray = blockIdx.x * kRaysPerBlock + wave_id();
    const bool live = ray < n;
    if (!live) ray = n - 1;
    for (int k = lane; k < nb; k += kWave) s_bins[k] = bins[(size_t)ray * nb + k];
    for (int j = lane; j < ns; j += kWave) s_u[j] = u[(size_t)ray * u_row_stride + j];
    build_cdf(weights + (size_t)ray * (nb - 1), nb, s_w, s_cdf, lane);
    search_right(s_cdf, nb, s_u, ns, s_ind, lane, false);
    block_sync();
    if (live) {
        for (int j = lane; j < ns; j += kWave) {
            const int ind = s_ind[j];
            samples[(size_t)ray * ns + j] = invert_cdf(s_cdf, s_bins, nb, s_u[j], ind);
            if (inds) inds[(size_t)ray * ns + j] = (int64_t)ind;
        }
        if (cdf_out)
            for (int k = lane; k < nb; k += kWave) cdf_out[(size_t)ray * nb + k] = s_cdf[k];
    }
}

__global__ __launch_bounds__(256) void coarse_sample_kernel(
    const float* __restrict__ rays, int ray_stride, const float* __restrict__ t_vals,
    const float* __restrict__ t_rand, float* __restrict__ z_out, float* __restrict__ pts, int n,
    int s, int lindisp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * s) return;
    const int ray = (int)(idx / s), i = (int)(idx - (long long)ray * s);
    const float* r = rays + (size_t)ray * ray_stride;
    const float z = ray::coarse_z(r[6], r[7], t_vals, i, s, lindisp, t_rand != nullptr, t_rand ? t_rand[idx] : 0.f);
    z_out[idx] = z;
    float* p = pts + idx * 3;
    p[0] = r[0] + r[3] * z;
    p[1] = r[1] + r[4] * z;
    p[2] = r[2] + r[5] * z;
}

__device__ __forceinline__ bool total_less(float a, int ia, float b, int ib) {
    // order used by the merge: numbers ascending, NaN last (as torch.sort), ties by position
    const bool an = a != a, bn = b != b;
    if (an || bn) return (!an && bn) || (an && bn && ia < ib);
    return (a < b) || (a == b && ia < ib);
}

__global__ __launch_bounds__(256) void fine_sample_kernel(
    const float* __restrict__ rays, int ray_stride, const float* __restrict__ z_c,
    const float* __restrict__ w_c, const float* __restrict__ u, int u_row_stride,
    float* __restrict__ z_f, float* __restrict__ pts_f, float* __restrict__ z_samples,
    float* __restrict__ z_std, int64_t* __restrict__ inds, float* __restrict__ cdf_out, int n,
    int sc, int sf, int lds_per_wave) {
    const int nb = sc - 1, tot = sc + sf;
    float* lds = dynamic_lds<float>() + (size_t)wave_id() * lds_per_wave;
    float* s_w = lds;
    float* s_cdf = s_w + nb;
    float* s_bins = s_cdf + nb;
    float* s_u = s_bins + nb;
    int* s_ind = reinterpret_cast<int*>(s_u + sf);
    float* s_all = reinterpret_cast<float*>(s_ind + sf);  // [tot] unsorted: z_c then samples
    float* s_sorted = s_all + tot;                        // [tot]
    const int lane = lane_id();
    int ray = blockIdx.x * kRaysPerBlock + wave_id();
    const bool live = ray < n;
    if (!live) ray = n - 1;
    const float* zc = z_c + (size_t)ray * sc;
    for (int k = lane; k < sc; k += kWave) s_all[k] = zc[k];
    for (int j = lane; j < sf; j += kWave) s_u[j] = u[(size_t)ray * u_row_stride + j];
    block_sync();
    for (int k = lane; k < nb; k += kWave) s_bins[k] = 0.5f * (s_all[k + 1] + s_all[k]);
    // weights[..., 1:-1]  (NeRF/render.py:270)
    build_cdf(w_c + (size_t)ray * sc + 1, nb, s_w, s_cdf, lane);
    search_right(s_cdf, nb, s_u, sf, s_ind, lane, false);
    block_sync();
    double part = 0.0;
    for (int j = lane; j < sf; j += kWave) {
        const float zs = invert_cdf(s_cdf, s_bins, nb, s_u[j], s_ind[j]);
        s_all[sc + j] = zs;
        part += (double)zs;
        if (live) {
            z_samples[(size_t)ray * sf + j] = zs;
            if (inds) inds[(size_t)ray * sf + j] = (int64_t)s_ind[j];
        }
    }
    if (live && cdf_out)
        for (int k = lane; k < nb; k += kWave) cdf_out[(size_t)ray * nb + k] = s_cdf[k];
    // population std of the new samples (two-pass, fp64)
    for (int o = 32; o > 0; o >>= 1) part += shfl_xor(part, o);
    const double mean = part / (double)sf;
    double var = 0.0;
    block_sync();
    for (int j = lane; j < sf; j += kWave) {
        const double dlt = (double)s_all[sc + j] - mean;
        var += dlt * dlt;
    }
    for (int o = 32; o > 0; o >>= 1) var += shfl_xor(var, o);
    if (live && lane == 0) z_std[ray] = (float)sqrt(var / (double)sf);
    // rank merge of the sc + sf depths (values only matter; equals torch.sort of the cat)
    for (int e = lane; e < tot; e += kWave) {
        const float v = s_all[e];
        int rank = 0;
#pragma unroll 4
        for (int j = 0; j < tot; ++j) rank += total_less(s_all[j], j, v, e) ? 1 : 0;
        s_sorted[rank] = v;
    }
    block_sync();
    if (live) {
        const float* r = rays + (size_t)ray * ray_stride;
        const float ox = r[0], oy = r[1], oz = r[2], dx = r[3], dy = r[4], dz = r[5];
        for (int e = lane; e < tot; e += kWave) {
            const float z = s_sorted[e];
            const size_t o = (size_t)ray * tot + e;
            z_f[o] = z;
            pts_f[o * 3 + 0] = ox + dx * z;
            pts_f[o * 3 + 1] = oy + dy * z;
            pts_f[o * 3 + 2] = oz + dz * z;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------- C ABI ------------
extern "C" int scnerf_abi_version(void) { return SCNERF_ABI_VERSION; }

extern "C" int scnerf_searchsorted(const float* a, const float* v, int64_t* out, int nrow,
                                   int nrow_a, int nrow_v, int na, int nv, int side_left,
                                   void* stream) {
    SCN_RETURN_IF(!a || !v || !out || nrow < 0 || na < 0 || nv < 0, SCN_EINVAL);
    SCN_RETURN_IF((nrow_a != 1 && nrow_a != nrow) || (nrow_v != 1 && nrow_v != nrow), SCN_EINVAL);
    if (nrow == 0 || nv == 0) return 0;
    const int per_wave = na + 2 * nv;
    const size_t lds = (size_t)per_wave * 4 * kRaysPerBlock;
    SCN_RETURN_IF(lds > 160 * 1024, SCN_ENOSUP);
    hipLaunchKernelGGL(searchsorted_kernel, dim3(scn_ceil_div(nrow, kRaysPerBlock)), dim3(256), lds,
                       (hipStream_t)stream, a, v, out, nrow, nrow_a, nrow_v, na, nv, side_left,
                       per_wave);
    return scn_launch_status();
}

extern "C" int scnerf_sample_pdf(const float* bins, const float* weights, const float* u,
                                 int u_row_stride, float* samples, int64_t* inds, float* cdf,
                                 int n, int nb, int ns, void* stream) {
    SCN_RETURN_IF(!bins || !weights || !u || !samples || n < 0 || nb < 2 || ns < 1, SCN_EINVAL);
    SCN_RETURN_IF(u_row_stride != 0 && u_row_stride != ns, SCN_EINVAL);
    if (n == 0) return 0;
    const int per_wave = 3 * nb + 2 * ns;
    const size_t lds = (size_t)per_wave * 4 * kRaysPerBlock;
    SCN_RETURN_IF(lds > 160 * 1024, SCN_ENOSUP);
    hipLaunchKernelGGL(sample_pdf_kernel, dim3(scn_ceil_div(n, kRaysPerBlock)), dim3(256), lds,
                       (hipStream_t)stream, bins, weights, u, u_row_stride, samples, inds, cdf, n,
                       nb, ns, per_wave);
    return scn_launch_status();
}

extern "C" int scnerf_coarse_sample(const float* rays, int ray_stride, const float* t_vals,
                                    const float* t_rand, float* z, float* pts, int n, int s,
                                    int lindisp, void* stream) {
    SCN_RETURN_IF(!rays || !t_vals || !z || !pts || n < 0 || s < 1 || ray_stride < 8, SCN_EINVAL);
    if (n == 0) return 0;
    hipLaunchKernelGGL(coarse_sample_kernel, dim3(scn_ceil_div((long long)n * s, 256)), dim3(256),
                       0, (hipStream_t)stream, rays, ray_stride, t_vals, t_rand, z, pts, n, s,
                       lindisp);
    return scn_launch_status();
}

extern "C" int scnerf_fine_sample(const float* rays, int ray_stride, const float* z_c,
                                  const float* w_c, const float* u, int u_row_stride, float* z_f,
                                  float* pts_f, float* z_samples, float* z_std, int64_t* inds,
                                  float* cdf, int n, int sc, int sf, void* stream) {
    SCN_RETURN_IF(!rays || !z_c || !w_c || !u || !z_f || !pts_f || !z_samples || !z_std, SCN_EINVAL);
    SCN_RETURN_IF(n < 0 || sc < 3 || sf < 1 || ray_stride < 8, SCN_EINVAL);
    SCN_RETURN_IF(u_row_stride != 0 && u_row_stride != sf, SCN_EINVAL);
    if (n == 0) return 0;
    const int per_wave = 3 * (sc - 1) + 2 * sf + 2 * (sc + sf);
    const size_t lds = (size_t)per_wave * 4 * kRaysPerBlock;
    SCN_RETURN_IF(lds > 160 * 1024, SCN_ENOSUP);
    hipLaunchKernelGGL(fine_sample_kernel, dim3(scn_ceil_div(n, kRaysPerBlock)), dim3(256), lds,
                       (hipStream_t)stream, rays, ray_stride, z_c, w_c, u, u_row_stride, z_f,
                       pts_f, z_samples, z_std, inds, cdf, n, sc, sf, per_wave);
    return scn_launch_status();
}
