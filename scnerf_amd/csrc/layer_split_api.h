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

}  // namespace lsp
}  // namespace scn
