// mlp_bwd_h3_api.h -- shared by the translation units of the resident data-gradient chain.
#pragma once
#include <hip/hip_runtime.h>

namespace scn {
namespace h3b {

// Where the weight-gradient GEMMs' chunk maxima of the dZ operands go (wgrad256_half.h): amax [8][n_chunks], job j =
// dZ of trunk layer j + 1 (j = 7: d feature); chunk = samples per weight-gradient workgroup.
struct ChunkMaxima { float* amax; int n_chunks; long chunk; };

int bwd_h3_pd3(const float* d_raw, const float* pts, const float* viewdirs, int vd_stride, int samples_per_ray,
               const float* wpacked_bwd, const short* stream_bwd, const float* scales, const float* save, float* grads,
               float* d_pts, float* d_views, long long n_samples, ChunkMaxima cm, hipStream_t st);
int bwd_h3_pd4(const float* d_raw, const float* pts, const float* viewdirs, int vd_stride, int samples_per_ray,
               const float* wpacked_bwd, const short* stream_bwd, const float* scales, const float* save, float* grads,
               float* d_pts, float* d_views, long long n_samples, ChunkMaxima cm, hipStream_t st);

}  // namespace h3b
}  // namespace scn
