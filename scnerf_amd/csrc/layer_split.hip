// layer_split.hip -- the 256-wide trunk layers of the network as per-layer GEMMs on the bf16 matrix pipe at fp32
// accuracy (layer_split.h), and the packing of their weights into bf16 planes.
//
// What it computes is nn.Linear (+ ReLU) of the reference network's trunk, layers 1 .. 7 and feature_linear
// (/root/reference NeRF/run_nerf_helpers.py:92-103, :105-128), layer 5 with the skip input [encoded point | h].
#include <cstdlib>
#include <scn_wave.h>

#include "launch.h"
#include "layer_split.h"
#include "mlp_common.h"
#include "scnerf_hip.h"

namespace {

using namespace scn;
using namespace scn::mlp;

// Plane buffer: the K slabs of layers l = 1 .. 8 (8 = feature_linear) one after the other; layer 5 has
// 16 + kEW / 16 slabs (h part first, then the encoded point in torch column order, zero beyond kInCh).
template <int PD>
__host__ __device__ constexpr int skip_slabs() { return Var<PD>::kEW / 16; }
template <int PD>
__host__ __device__ constexpr int slab_offset(int l) {
    return l <= 5 ? (l - 1) * 16 : 16 * (l - 1) + skip_slabs<PD>();
}
template <int PD>
__host__ __device__ constexpr int fwd_slabs() { return slab_offset<PD>(8) + 16; }
// then the TRANSPOSED weights of the data-gradient chain, 16 slabs each: entry 0 = feature_linear^T, entry e = 1 .. 7
// = layer (8 - e)^T (layer 5: its h columns; the encoded-point columns stay with the fused kernel's last stage)
template <int PD>
__host__ __device__ constexpr int total_slabs() { return fwd_slabs<PD>() + 8 * 16; }

// one thread per (slab, feature tile T, lane, element e): the three planes of one weight
template <int PD>
__global__ __launch_bounds__(256) void pack_planes_kernel(const float* __restrict__ params, short* __restrict__ out) {
    using V = Var<PD>;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)total_slabs<PD>() * 8 * 64 * 8) return;
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63), T = (int)((idx >> 9) & 7);
    const int slab = (int)(idx >> 12);
    if (slab >= fwd_slabs<PD>()) {
        const int entry = (slab - fwd_slabs<PD>()) >> 4, s = (slab - fwd_slabs<PD>()) & 15;
        const int row = 32 * T + (lane & 31);                   // output of the transposed layer = input feature k
        const int col = 16 * s + 8 * (lane >> 5) + e;           // contraction = output feature n of the layer
        const int l = 8 - entry;
        const float w = entry == 0 ? params[V::kWF + col * 256 + row]
                      : l == 5   ? params[V::trunk_w(5) + col * V::kSkipLd + V::kInCh + row]
                                 : params[V::trunk_w(l) + col * 256 + row];
        const unsigned u = __float_as_uint(w);
        const float d1 = w - __uint_as_float(u & 0xffff0000u);
        const unsigned u1 = __float_as_uint(d1);
        const float d2 = d1 - __uint_as_float(u1 & 0xffff0000u);
        const unsigned u2 = __float_as_uint(d2);
        const long base = (((long)slab * 3) * 8 + T) * 64 * 8 + lane * 8 + e;
        out[base] = (short)(u >> 16);
        out[base + 8 * 64 * 8] = (short)(u1 >> 16);
        out[base + 2 * 8 * 64 * 8] = (short)(u2 >> 16);
        return;
    }
    int l = 1;
#pragma unroll
    for (int c = 2; c <= 8; ++c)
        if (slab >= slab_offset<PD>(c)) l = c;
    const int s = slab - slab_offset<PD>(l);
    const int n = 32 * T + (lane & 31);
    const int k = 16 * (s & 15) + 8 * (lane >> 5) + e;        // column inside the part
    float w;
    if (l == 8) {
        w = params[V::kWF + n * 256 + k];
    } else if (l == 5) {
        // torch column order of the skip layer's input: [encoded point (kInCh) | h (256)]
        if (s < 16) w = params[V::trunk_w(5) + n * V::kSkipLd + V::kInCh + k];
        else {
            const int c = 16 * (s - 16) + 8 * (lane >> 5) + e;
            w = c < V::kInCh ? params[V::trunk_w(5) + n * V::kSkipLd + c] : 0.f;
        }
    } else {
        w = params[V::trunk_w(l) + n * 256 + k];
    }
    const unsigned u = __float_as_uint(w);
    const float d1 = w - __uint_as_float(u & 0xffff0000u);
    const unsigned u1 = __float_as_uint(d1);
    const float d2 = d1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned u2 = __float_as_uint(d2);
    const long base = (((long)slab * 3) * 8 + T) * 64 * 8 + lane * 8 + e;
    out[base] = (short)(u >> 16);
    out[base + 8 * 64 * 8] = (short)(u1 >> 16);
    out[base + 2 * 8 * 64 * 8] = (short)(u2 >> 16);
}

}  // namespace

namespace scn {
namespace lsp {

// persistent workgroups per launch (one per CU of the MI355X; scnerf_layer_split_workgroups lowers it, e.g. to give
// every workgroup several blocks of a small problem)
static int& max_workgroups() {
    static int g = 256;
    return g;
}

// blocks per group of a chain (layer_split.h); SCNERF_LAYER_GROUP overrides (experiments)
static int chain_group() {
    static const int g = [] {
        const char* e = getenv("SCNERF_LAYER_GROUP");
        const int v = e ? atoi(e) : 0;
        return v >= 2 ? v : 2;
    }();
    return g;
}

static int launch(const Args& a, hipStream_t stream) {
    const long n_blocks = (a.Ppad / 32 + 7) / 8;
    const long cap = max_workgroups();
    const unsigned G = (unsigned)(n_blocks < cap ? n_blocks : cap);
    if (a.n_layers > 1) {       // a chain: what a layer stores is read back two blocks later -- default cache policy
        SCN_LDS_OPT_IN((layer_split_kernel<kPlainStore>), kLdsBytes);
        hipLaunchKernelGGL((layer_split_kernel<kPlainStore>), dim3(G), dim3(kThreads), kLdsBytes, stream, a);
    } else {
        SCN_LDS_OPT_IN((layer_split_kernel<0>), kLdsBytes);
        hipLaunchKernelGGL((layer_split_kernel<0>), dim3(G), dim3(kThreads), kLdsBytes, stream, a);
    }
    return scn_launch_status();
}

// A chain goes out as ONE launch when every workgroup owns at least two 256-sample blocks (the pipeline then fetches
// the next layer's first block long after it was stored); otherwise layer by layer.
static int launch_chain(const Args& chain, hipStream_t stream) {
    const long n_blocks = (chain.Ppad / 32 + 7) / 8;
    if (chain.n_layers == 1 || n_blocks >= 2L * max_workgroups()) return launch(chain, stream);
    for (int l = 0; l < chain.n_layers; ++l) {
        Args one = chain;
        one.layer[0] = chain.layer[l];
        one.n_layers = 1;
        const int rc = launch(one, stream);
        if (rc) return rc;
    }
    return 0;
}

template <int PD>
static Layer forward_layer(int l, const short* planes, const float* bias_table, const float* act_in, const float* epts,
                           float* act_out, unsigned* mask) {
    Layer L;
    L.X = act_in;
    L.X2 = l == 5 ? epts : act_in;          // (a valid pointer either way: the select in load_x forms both addresses)
    L.n_k = l == 5 ? 16 + skip_slabs<PD>() : 16;
    L.relu = l < 8;
    L.W = planes + (long)slab_offset<PD>(l) * kSlabShorts;
    L.bias = bias_table;
    L.Z = act_out;
    L.mask = mask;
    L.mask_in = nullptr;
    L.vec = nullptr;
    return L;
}

template <int PD>
static Layer backward_layer(int entry, const short* planes, const float* alpha_table, const float* grad_in,
                            float* grad_out, const unsigned* mask_in, const float* vec) {
    Layer L;
    L.X = grad_in;
    L.X2 = grad_in;
    L.n_k = 16;
    L.relu = 0;
    L.W = planes + (long)(fwd_slabs<PD>() + 16 * entry) * kSlabShorts;
    L.bias = alpha_table;                    // read into LDS whether used or not: must be a valid table
    L.Z = grad_out;
    L.mask = nullptr;
    L.mask_in = mask_in;
    L.vec = entry == 0 ? vec : nullptr;
    return L;
}

static Args chain_header(int n_layers, int x2_ld, long Ppad, int mode, int vec_stride, long n_vec) {
    Args a;
    a.n_layers = n_layers;
    a.clock_probe = nullptr;
    a.x_scale = a.out_scale = 1.f;
    a.group = chain_group();
    a.x2_ld = x2_ld;
    a.Ppad = Ppad;
    a.mode = mode;
    a.vec_stride = vec_stride;
    a.n_vec = n_vec;
    return a;
}

// layer l = 1 .. 8 of the network out of the plane buffer; act_in / act_out tile-native sections
template <int PD>
int launch_network_layer(int l, const short* planes, const float* bias_table, const float* act_in, const float* epts,
                         float* act_out, unsigned* mask, long Ppad, hipStream_t stream) {
    Args a = chain_header(1, Var<PD>::kEW, Ppad, 0, 0, 0);
    a.layer[0] = forward_layer<PD>(l, planes, bias_table, act_in, epts, act_out, mask);
    return launch_chain(a, stream);
}

// data-gradient layer `entry` (0: feature_linear^T + the density head's rank-1 term; e = 1 .. 7: layer (8 - e)^T):
// grad_out = gate(mask_in, W^T grad_in [+ alpha_table[n] * vec[p]])
template <int PD>
int launch_network_layer_bwd(int entry, const short* planes, const float* alpha_table, const float* grad_in,
                             float* grad_out, const unsigned* mask_in, const float* vec, int vec_stride, long n_vec,
                             long Ppad, hipStream_t stream) {
    Args a = chain_header(1, Var<PD>::kEW, Ppad, 1, vec_stride, n_vec);
    a.layer[0] = backward_layer<PD>(entry, planes, alpha_table, grad_in, grad_out, mask_in, vec);
    return launch_chain(a, stream);
}

// layers 1 .. 8 over the training workspace `save` (sections of mlp_common.h) as one chain
template <int PD>
int launch_network_chain_fwd(const short* planes, const float* wpacked, float* save, long P, hipStream_t stream) {
    using V = Var<PD>;
    const long Ppad = padded_samples(P);
    unsigned* masks = reinterpret_cast<unsigned*>(save + (long)V::kSavePerSample * Ppad);
    const float* epts = save + (long)kSaveEpts * Ppad;
    Args a = chain_header(8, V::kEW, Ppad, 0, 0, 0);
    for (int l = 1; l <= 8; ++l)
        a.layer[l - 1] = forward_layer<PD>(l, planes, wpacked + (l < 8 ? V::kFwdBias + 256 * l : V::kFwdBiasF),
                                           save + (long)(kSaveAct + 256 * (l - 1)) * Ppad, epts,
                                           save + (long)(l < 8 ? kSaveAct + 256 * l : kSaveFeat) * Ppad,
                                           l < 8 ? masks + (long)l * (Ppad / 32) * 256 : nullptr);
    return launch_chain(a, stream);
}

// entries 0 .. 7 of the data-gradient chain over the gradient workspace: d feature -> dZ_7 -> ... -> dZ_0
template <int PD>
int launch_network_chain_bwd(const short* planes, const float* wpacked_bwd, const float* save, float* grads,
                             const float* d_raw, long P, hipStream_t stream) {
    using V = Var<PD>;
    const long Ppad = padded_samples(P);
    const unsigned* masks = reinterpret_cast<const unsigned*>(save + (long)V::kSavePerSample * Ppad);
    const float* alpha = wpacked_bwd + V::kBwdAlphaW;
    Args a = chain_header(8, V::kEW, Ppad, 1, 4, P);
    for (int e = 0; e < 8; ++e)
        a.layer[e] = backward_layer<PD>(e, planes, alpha, grads + (long)(e == 0 ? kGradDfeat : kGradDz + (8 - e) * 256) * Ppad,
                                        grads + (long)(kGradDz + (7 - e) * 256) * Ppad,
                                        masks + (long)(7 - e) * (Ppad / 32) * 256, d_raw + 3);
    return launch_chain(a, stream);
}

template int launch_network_layer<3>(int, const short*, const float*, const float*, const float*, float*, unsigned*, long, hipStream_t);
template int launch_network_layer<4>(int, const short*, const float*, const float*, const float*, float*, unsigned*, long, hipStream_t);
template int launch_network_layer_bwd<3>(int, const short*, const float*, const float*, float*, const unsigned*, const float*, int, long, long, hipStream_t);
template int launch_network_layer_bwd<4>(int, const short*, const float*, const float*, float*, const unsigned*, const float*, int, long, long, hipStream_t);
template int launch_network_chain_fwd<3>(const short*, const float*, float*, long, hipStream_t);
template int launch_network_chain_fwd<4>(const short*, const float*, float*, long, hipStream_t);
template int launch_network_chain_bwd<3>(const short*, const float*, const float*, float*, const float*, long, hipStream_t);
template int launch_network_chain_bwd<4>(const short*, const float*, const float*, float*, const float*, long, hipStream_t);

}  // namespace lsp
}  // namespace scn

extern "C" int scnerf_layer_split_workgroups(int n) {
    if (n >= 1 && n <= 1024) scn::lsp::max_workgroups() = n;
    return scn::lsp::max_workgroups();
}

extern "C" int scnerf_layer_split_chain_fwd(int pt_dims, const short* planes, const float* wpacked, float* save,
                                            long long n_samples, void* stream) {
    SCN_RETURN_IF(!planes || !wpacked || !save || n_samples < 0 || (pt_dims != 3 && pt_dims != 4), SCN_EINVAL);
    if (n_samples == 0) return 0;
    return pt_dims == 3 ? scn::lsp::launch_network_chain_fwd<3>(planes, wpacked, save, (long)n_samples, (hipStream_t)stream)
                        : scn::lsp::launch_network_chain_fwd<4>(planes, wpacked, save, (long)n_samples, (hipStream_t)stream);
}

extern "C" int scnerf_layer_split_chain_bwd(int pt_dims, const short* planes, const float* wpacked_bwd, const float* save,
                                            float* grads, const float* d_raw, long long n_samples, void* stream) {
    SCN_RETURN_IF(!planes || !wpacked_bwd || !save || !grads || !d_raw || n_samples < 0 || (pt_dims != 3 && pt_dims != 4), SCN_EINVAL);
    if (n_samples == 0) return 0;
    return pt_dims == 3 ? scn::lsp::launch_network_chain_bwd<3>(planes, wpacked_bwd, save, grads, d_raw, (long)n_samples, (hipStream_t)stream)
                        : scn::lsp::launch_network_chain_bwd<4>(planes, wpacked_bwd, save, grads, d_raw, (long)n_samples, (hipStream_t)stream);
}

extern "C" long long scnerf_split_planes_shorts(int pt_dims) {
    if (pt_dims != 3 && pt_dims != 4) return -1;
    return (long long)(pt_dims == 3 ? total_slabs<3>() : total_slabs<4>()) * scn::lsp::kSlabShorts;
}

extern "C" int scnerf_pack_split_planes(int pt_dims, const float* flat_params, short* planes, void* stream) {
    SCN_RETURN_IF(!flat_params || !planes || (pt_dims != 3 && pt_dims != 4), SCN_EINVAL);
    const long n = (long)(pt_dims == 3 ? total_slabs<3>() : total_slabs<4>()) * 8 * 64 * 8;
    if (pt_dims == 3)
        hipLaunchKernelGGL(pack_planes_kernel<3>, dim3(scn_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, flat_params, planes);
    else
        hipLaunchKernelGGL(pack_planes_kernel<4>, dim3(scn_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, flat_params, planes);
    return scn_launch_status();
}

extern "C" int scnerf_layer_split(int pt_dims, int layer, const short* planes, const float* bias_table, const float* act_in,
                                  const float* epts, float* act_out, unsigned* mask, long long n_samples, void* stream) {
    SCN_RETURN_IF(!planes || !bias_table || !act_in || !act_out || n_samples < 0, SCN_EINVAL);
    SCN_RETURN_IF((pt_dims != 3 && pt_dims != 4) || layer < 1 || layer > 8 || (layer == 5 && !epts), SCN_EINVAL);
    if (n_samples == 0) return 0;
    const long Ppad = padded_samples(n_samples);
    hipStream_t st = (hipStream_t)stream;
    return pt_dims == 3 ? scn::lsp::launch_network_layer<3>(layer, planes, bias_table, act_in, epts, act_out, mask, Ppad, st)
                        : scn::lsp::launch_network_layer<4>(layer, planes, bias_table, act_in, epts, act_out, mask, Ppad, st);
}

extern "C" int scnerf_layer_split_bwd(int pt_dims, int entry, const short* planes, const float* alpha_table,
                                      const float* grad_in, float* grad_out, const unsigned* mask_in, const float* d_raw,
                                      long long n_samples, void* stream) {
    SCN_RETURN_IF(!planes || !alpha_table || !grad_in || !grad_out || !mask_in || !d_raw || n_samples < 0, SCN_EINVAL);
    SCN_RETURN_IF((pt_dims != 3 && pt_dims != 4) || entry < 0 || entry > 7, SCN_EINVAL);
    if (n_samples == 0) return 0;
    const long Ppad = padded_samples(n_samples);
    hipStream_t st = (hipStream_t)stream;
    return pt_dims == 3 ? scn::lsp::launch_network_layer_bwd<3>(entry, planes, alpha_table, grad_in, grad_out, mask_in, d_raw + 3, 4, (long)n_samples, Ppad, st)
                        : scn::lsp::launch_network_layer_bwd<4>(entry, planes, alpha_table, grad_in, grad_out, mask_in, d_raw + 3, 4, (long)n_samples, Ppad, st);
}
