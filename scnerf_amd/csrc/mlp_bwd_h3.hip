// mlp_bwd_h3.hip -- the data-gradient chain of the NeRF MLP in the "resident" arithmetic (mlp_h3.h): d raw ->
// rgb^T -> views^T -> feature^T (+ the density head's rank-1 term) -> trunk layers 7^T .. 1^T with the ReLU gates ->
// layer 0^T and the skip columns -> the encoding's gradient -> d pts, d viewdirs, as ONE launch on three fp16
// products per product.  The gradient w.r.t. a layer's output lives in registers as cut fp16 planes and is the B
// operand of the transposed layer; every dZ the weight-gradient GEMMs need is written once (tile-native sections of
// mlp_common.h, as mlp_bwd.hip) and never read back here.
//
// Gradient of: NeRF.forward + Embedder + run_network
//   /root/reference NeRF/run_nerf_helpers.py:105-128, :24-72, NeRF/create_nerf.py:18-32 (what autograd derives).
#include <scn_wave.h>

#include "launch.h"
#include "mlp_bwd_h3_api.h"
#include "scnerf_hip.h"

extern "C" int scnerf_mlp_bwd_h3(int pt_dims, const float* d_raw, const float* pts, const float* viewdirs, int vd_stride,
                                 int samples_per_ray, const float* wpacked_bwd, const short* stream_bwd, const float* scales,
                                 const float* save, float* grads, float* d_pts, float* d_views, long long n_samples,
                                 float* chunk_amax, int n_chunks, long long chunk_samples, void* stream) {
    SCN_RETURN_IF(!d_raw || !pts || !viewdirs || !wpacked_bwd || !stream_bwd || !scales || !save || !grads || !d_pts || !d_views, SCN_EINVAL);
    SCN_RETURN_IF(samples_per_ray < 1 || vd_stride < 3 || n_samples < 0 || (pt_dims != 3 && pt_dims != 4), SCN_EINVAL);
    SCN_RETURN_IF(n_samples >= (1LL << 31), SCN_ENOSUP);       // (the kernel indexes samples with 31 bits)
    SCN_RETURN_IF(chunk_amax && (n_chunks < 1 || chunk_samples < 32 || chunk_samples % 32), SCN_EINVAL);
    if (n_samples == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const scn::h3b::ChunkMaxima cm{chunk_amax, n_chunks, (long)chunk_samples};
    return pt_dims == 3 ? scn::h3b::bwd_h3_pd3(d_raw, pts, viewdirs, vd_stride, samples_per_ray, wpacked_bwd, stream_bwd, scales, save, grads, d_pts, d_views, n_samples, cm, st)
                        : scn::h3b::bwd_h3_pd4(d_raw, pts, viewdirs, vd_stride, samples_per_ray, wpacked_bwd, stream_bwd, scales, save, grads, d_pts, d_views, n_samples, cm, st);
}
