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
#include "ray_sample.h"
#include "ray_stage.h"
#include "scnerf_hip.h"

namespace {

using namespace scn;

constexpr int kRaysPerBlock = 4;  // 4 waves / 256 threads

using namespace scn::ray;

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

__global__ __launch_bounds__(256) void fine_sample_kernel(
    const float* __restrict__ rays, int ray_stride, const float* __restrict__ z_c,
    const float* __restrict__ w_c, const float* __restrict__ u, int u_row_stride,
    float* __restrict__ z_f, float* __restrict__ pts_f, float* __restrict__ z_samples,
    float* __restrict__ z_std, int64_t* __restrict__ inds, float* __restrict__ cdf_out, int n,
    int sc, int sf, int lds_per_wave) {
    const int tot = sc + sf;
    float* lds = dynamic_lds<float>() + (size_t)wave_id() * lds_per_wave;
    int ray = blockIdx.x * kRaysPerBlock + wave_id();
    const bool live = ray < n;
    if (!live) ray = n - 1;
    ray::fine_sample_ray(rays + (size_t)ray * ray_stride, z_c + (size_t)ray * sc, w_c + (size_t)ray * sc,
                         u + (size_t)ray * u_row_stride, sc, sf, lds, lane_id(), live, z_f + (size_t)ray * tot,
                         pts_f + (size_t)ray * tot * 3, z_samples + (size_t)ray * sf, z_std + ray,
                         inds ? inds + (size_t)ray * sf : nullptr, cdf_out ? cdf_out + (size_t)ray * (sc - 1) : nullptr);
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
    const int per_wave = ray::fine_sample_lds_floats(sc, sf);
    const size_t lds = (size_t)per_wave * 4 * kRaysPerBlock;
    SCN_RETURN_IF(lds > 160 * 1024, SCN_ENOSUP);
    hipLaunchKernelGGL(fine_sample_kernel, dim3(scn_ceil_div(n, kRaysPerBlock)), dim3(256), lds,
                       (hipStream_t)stream, rays, ray_stride, z_c, w_c, u, u_row_stride, z_f,
                       pts_f, z_samples, z_std, inds, cdf, n, sc, sf, per_wave);
    return scn_launch_status();
}
