// mlp_fwd_h3_coarse.hip -- one instantiation group of the resident forward kernel (mlp_fwd_h3_kernel.h).
#include "mlp_fwd_h3_kernel.h"

namespace scn {
namespace h3f {

int fwd_h3_coarse(const CoarseStage& cs, const float* rays, int ray_stride, const float* wpacked, const short* stream_fwd,
                  const float* scales, float* raw, float* save, ChunkMaxima cm, hipStream_t st) {
    return save ? launch_coarse_h3<true>(cs, rays, ray_stride, wpacked, stream_fwd, scales, raw, save, cm, st)
                : launch_coarse_h3<false>(cs, rays, ray_stride, wpacked, stream_fwd, scales, raw, save, cm, st);
}

}  // namespace h3f
}  // namespace scn
