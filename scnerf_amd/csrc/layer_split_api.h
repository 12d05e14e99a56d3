// layer_split_api.h -- host-side launch interface of the per-layer split-arithmetic GEMMs (layer_split.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace scn {
namespace lsp {

// layer l = 1 .. 8 (8 = feature_linear) of network variant PD out of the plane buffer of scnerf_pack_split_planes:
// act_out = act(W_l [act_in | epts] + b); sections are tile-native of width 256 over Ppad samples, epts row-major
// [Ppad][64 | 128] (used by layer 5), bias_table the layer's lane-vector table, mask its ReLU bit section or nullptr.
template <int PD>
int launch_network_layer(int l, const short* planes, const float* bias_table, const float* act_in, const float* epts,
                         float* act_out, unsigned* mask, long Ppad, hipStream_t stream);

// data-gradient layer `entry` (0: feature_linear^T + alpha_table[n] * vec[p * vec_stride], p < n_vec; e = 1 .. 7:
// trunk layer (8 - e)^T): grad_out = gate(mask_in, W^T grad_in [+ rank-1 term]); alpha_table must be a valid
// 256-float lane-vector table in either case
template <int PD>
int launch_network_layer_bwd(int entry, const short* planes, const float* alpha_table, const float* grad_in,
                             float* grad_out, const unsigned* mask_in, const float* vec, int vec_stride, long n_vec,
                             long Ppad, hipStream_t stream);

// layers 1 .. 8 of the training forward over the workspace `save` / entries 0 .. 7 of the data-gradient chain over
// `grads`, as ONE launch when every workgroup owns at least two 256-sample blocks (else layer by layer)
// amax: nullptr (all layers on six bf16 products), or the workspace of scnerf_layer_amax_floats: the layers fed with
// per-sample maxima then run on three fp16 products (layer_split.h)
template <int PD>
int launch_network_chain_fwd(const short* planes, const float* wpacked, float* save, float* amax, long P, hipStream_t stream);
template <int PD>
int launch_network_chain_bwd(const short* planes, const float* wpacked_bwd, const float* save, float* grads,
                             const float* d_raw, float* amax, long P, hipStream_t stream);

}  // namespace lsp
}  // namespace scn
