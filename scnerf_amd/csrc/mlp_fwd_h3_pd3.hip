// mlp_fwd_h3_pd3.hip -- one instantiation group of the resident forward kernel (mlp_fwd_h3_kernel.h).
#include "mlp_fwd_h3_kernel.h"

namespace scn {
namespace h3f {

int fwd_h3_pd3(const float* pts, const float* viewdirs, int vd_stride, int samples_per_ray, const float* wpacked,
               const short* stream_fwd, const float* scales, float* raw, float* save, long long n_samples, ChunkMaxima cm,
               hipStream_t st) {
    return save ? launch_fwd_h3<3, true>(pts, viewdirs, vd_stride, samples_per_ray, wpacked, stream_fwd, scales, raw, save, n_samples, cm, st)
                : launch_fwd_h3<3, false>(pts, viewdirs, vd_stride, samples_per_ray, wpacked, stream_fwd, scales, raw, save, n_samples, cm, st);
}

}  // namespace h3f
}  // namespace scn
