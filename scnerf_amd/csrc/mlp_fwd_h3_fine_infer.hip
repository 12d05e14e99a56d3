// mlp_fwd_h3_fine_infer.hip -- one instantiation of the resident forward kernel (mlp_fwd_h3_kernel.h): the fused fine stage,
// inference (nothing saved for a backward pass).
#include "mlp_fwd_h3_kernel.h"

namespace scn {
namespace h3f {

int fwd_h3_fine_infer(const FineStage& fs, const float* wpacked, const short* stream_fwd, const float* scales, float* raw,
                      ChunkMaxima cm, hipStream_t st) {
    return launch_fine_h3<false>(fs, wpacked, stream_fwd, scales, raw, nullptr, cm, st);
}

}  // namespace h3f
}  // namespace scn
