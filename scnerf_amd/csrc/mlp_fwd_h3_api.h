// mlp_fwd_h3_api.h -- what the translation units of the resident forward share: the argument structs and the
// per-instantiation-group launchers (defined in mlp_fwd_h3_pd3.hip, mlp_fwd_h3_pd4.hip, mlp_fwd_h3_coarse.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace scn {
namespace h3f {

// Where the weight-gradient GEMMs' chunk maxima go (wgrad256_half.h): amax [8][n_chunks], job j = the GEMM whose X
// operand is the input of trunk layer j + 1 (j = 7: feature_linear); chunk = samples per weight-gradient workgroup.
struct ChunkMaxima { float* amax; int n_chunks; long chunk; };

struct CoarseStage {
    const float* rays; int ray_stride; int n_rays;
    const float* t_vals; const float* t_rand; int lindisp;
    float* z; float* pts;
    const float* noise; int white_bkgd;
    float* rgb; float* disp; float* acc; float* depth; float* weights;
};
constexpr int kCoarseSamples = 64;

// pt_dims = 3 / 4, fine stage; train = (save != nullptr)
int fwd_h3_pd3(const float* pts, const float* viewdirs, int vd_stride, int samples_per_ray, const float* wpacked,
               const short* stream_fwd, const float* scales, float* raw, float* save, long long n_samples, ChunkMaxima cm,
               hipStream_t st);
int fwd_h3_pd4(const float* pts, const float* viewdirs, int vd_stride, int samples_per_ray, const float* wpacked,
               const short* stream_fwd, const float* scales, float* raw, float* save, long long n_samples, ChunkMaxima cm,
               hipStream_t st);
// the fused coarse stage (sampling + network + compositing)
int fwd_h3_coarse(const CoarseStage& cs, const float* rays, int ray_stride, const float* wpacked, const short* stream_fwd,
                  const float* scales, float* raw, float* save, ChunkMaxima cm, hipStream_t st);

}  // namespace h3f
}  // namespace scn
