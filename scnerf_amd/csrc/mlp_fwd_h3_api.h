// mlp_fwd_h3_api.h -- what the translation units of the resident forward share: the argument structs and the
// per-instantiation-group launchers (defined in mlp_fwd_h3_pd3.hip, mlp_fwd_h3_pd4.hip, mlp_fwd_h3_coarse.hip).
#pragma once
#include <cstdint>

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

// The fused fine stage (mlp_fwd_h3_kernel.h, STAGE kFine): what the sampler reads (the coarse stage's depths and weights,
// the uniforms), what it writes for the data-gradient pass (merged depths, their points), and the compositing's outputs.
struct FineStage {
    const float* rays; int ray_stride; int n_rays;
    const float* z_c; const float* w_c;                       // [n_rays, 64]
    const float* u; int u_row_stride; int n_importance;       // [n_rays, n_importance] (stride 0: one row for all)
    float* z_f; float* pts_f; float* z_samples; float* z_std; int64_t* inds; float* cdf;
    const float* noise; int white_bkgd;                       // density noise [n_rays, 64 + n_importance] or nullptr
    float* rgb; float* disp; float* acc; float* depth; float* weights;
    int rays_per_block, tiles;                                 // rays_per_block x (64 + n_importance) = tiles x 128
};

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
// the fused fine stage (sampler + merge + network + compositing); 64 + n_importance in {128, 192, 256}
int fwd_h3_fine_train(const FineStage& fs, const float* wpacked, const short* stream_fwd, const float* scales, float* raw,
                      float* save, ChunkMaxima cm, hipStream_t st);
int fwd_h3_fine_infer(const FineStage& fs, const float* wpacked, const short* stream_fwd, const float* scales, float* raw,
                      ChunkMaxima cm, hipStream_t st);

}  // namespace h3f
}  // namespace scn
