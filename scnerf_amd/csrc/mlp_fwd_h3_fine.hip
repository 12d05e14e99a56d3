// mlp_fwd_h3_fine.hip -- one instantiation of the resident forward kernel (mlp_fwd_h3_kernel.h): the fused fine stage, training.
// (Its own translation unit, and the inference twin another: the three passes are straight-line code, ~3 minutes to compile.)
#include "mlp_fwd_h3_kernel.h"

namespace scn {
namespace h3f {

int fwd_h3_fine_train(const FineStage& fs, const float* wpacked, const short* stream_fwd, const float* scales, float* raw,
                      float* save, ChunkMaxima cm, hipStream_t st) {
    return launch_fine_h3<true>(fs, wpacked, stream_fwd, scales, raw, save, cm, st);
}

}  // namespace h3f
}  // namespace scn
