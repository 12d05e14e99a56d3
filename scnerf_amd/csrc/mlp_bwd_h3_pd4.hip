// mlp_bwd_h3_pd4.hip -- one network variant of the resident data-gradient kernel (mlp_bwd_h3_kernel.h).
#include "mlp_bwd_h3_kernel.h"

namespace scn {
namespace h3b {

int bwd_h3_pd4(const float* d_raw, const float* pts, const float* viewdirs, int vd_stride, int samples_per_ray,
               const float* wpacked_bwd, const short* stream_bwd, const float* scales, const float* save, float* grads,
               float* d_pts, float* d_views, long long n_samples, ChunkMaxima cm, hipStream_t st) {
    return launch_bwd_h3<4>(d_raw, pts, viewdirs, vd_stride, samples_per_ray, wpacked_bwd, stream_bwd, scales, save, grads, d_pts,
                            d_views, n_samples, cm, st);
}

}  // namespace h3b
}  // namespace scn
