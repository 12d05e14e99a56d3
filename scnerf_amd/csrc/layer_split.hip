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

// the weight a thread (slab, feature tile T, lane, element e) of the plane image encodes, and the network layer
// (1 .. 8) it belongs to
template <int PD>
__device__ __forceinline__ float plane_weight(const float* __restrict__ params, int slab, int T, int lane, int e, int* layer) {
    using V = Var<PD>;
    if (slab >= fwd_slabs<PD>()) {
        const int entry = (slab - fwd_slabs<PD>()) >> 4, s = (slab - fwd_slabs<PD>()) & 15;
        const int row = 32 * T + (lane & 31);                   // output of the transposed layer = input feature k
        const int col = 16 * s + 8 * (lane >> 5) + e;           // contraction = output feature n of the layer
        const int l = 8 - entry;
        *layer = l;
        return entry == 0 ? params[V::kWF + col * 256 + row]
             : l == 5   ? params[V::trunk_w(5) + col * V::kSkipLd + V::kInCh + row]
                        : params[V::trunk_w(l) + col * 256 + row];
    }
    int l = 1;
#pragma unroll
    for (int c = 2; c <= 8; ++c)
        if (slab >= slab_offset<PD>(c)) l = c;
    *layer = l;
    const int s = slab - slab_offset<PD>(l);
    const int n = 32 * T + (lane & 31);
    const int k = 16 * (s & 15) + 8 * (lane >> 5) + e;        // column inside the part
    if (l == 8) return params[V::kWF + n * 256 + k];
    if (l == 5) {
        // torch column order of the skip layer's input: [encoded point (kInCh) | h (256)]
        if (s < 16) return params[V::trunk_w(5) + n * V::kSkipLd + V::kInCh + k];
        const int c = 16 * (s - 16) + 8 * (lane >> 5) + e;
        return c < V::kInCh ? params[V::trunk_w(5) + n * V::kSkipLd + c] : 0.f;
    }
    return params[V::trunk_w(l) + n * 256 + k];
}

// one thread per (slab, feature tile T, lane, element e).  HALF = false: the three bf16 planes of the weight (exact
// cut).  HALF = true: two fp16 planes of weight x scale[layer] (the third plane stays empty), scale = the power of
// two weight_scale_kernel chose for the layer.
template <int PD, bool HALF>
__global__ __launch_bounds__(256) void pack_planes_kernel(const float* __restrict__ params, short* __restrict__ out,
                                                          const float* __restrict__ scale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)total_slabs<PD>() * 8 * 64 * 8) return;
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63), T = (int)((idx >> 9) & 7);
    const int slab = (int)(idx >> 12);
    int layer;
    const float w = plane_weight<PD>(params, slab, T, lane, e, &layer);
    const long base = (((long)slab * 3) * 8 + T) * 64 * 8 + lane * 8 + e;
    if constexpr (HALF) {
        const float x = w * scale[layer - 1];
        const unsigned h = f16_bits(x);
        out[base] = (short)h;
        out[base + 8 * 64 * 8] = (short)f16_bits(x - f16_value(h));
        out[base + 2 * 8 * 64 * 8] = 0;
    } else {
        const unsigned u = __float_as_uint(w);
        const float d1 = w - __uint_as_float(u & 0xffff0000u);
        const unsigned u1 = __float_as_uint(d1);
        const float d2 = d1 - __uint_as_float(u1 & 0xffff0000u);
        const unsigned u2 = __float_as_uint(d2);
        out[base] = (short)(u >> 16);
        out[base + 8 * 64 * 8] = (short)(u1 >> 16);
        out[base + 2 * 8 * 64 * 8] = (short)(u2 >> 16);
    }
}

// one workgroup per layer l = 1 .. 8: the power of two 2^k with |w| 2^k < 2^13 for every weight of the layer
// -> tail[l - 1] = 1 / 2^k (what the layer kernel multiplies its results by), tail[8 + l - 1] = 2^k
template <int PD>
__global__ __launch_bounds__(256) void weight_scale_kernel(const float* __restrict__ params, float* __restrict__ tail) {
    using V = Var<PD>;
    float* red = dynamic_lds<float>();                      // 256 floats
    const int l = blockIdx.x + 1;
    const float* w = params + (l == 8 ? V::kWF : V::trunk_w(l));
    const int n = 256 * (l == 5 ? V::kSkipLd : 256);
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    block_sync();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        block_sync();
    }
    if (threadIdx.x == 0) {
        const unsigned e = (__float_as_uint(red[0]) >> 23) & 0xffu;
        const unsigned bits = (266u - (e < 13u ? 13u : e)) << 23;
        tail[8 + l - 1] = __uint_as_float(bits);
        tail[l - 1] = __uint_as_float(0x7f000000u - bits);
    }
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

template <int FLAGS>
static int launch_as(const Args& a, unsigned G, hipStream_t stream) {
    SCN_LDS_OPT_IN((layer_split_kernel<FLAGS>), kLdsBytes);
    hipLaunchKernelGGL((layer_split_kernel<FLAGS>), dim3(G), dim3(kThreads), kLdsBytes, stream, a);
    return scn_launch_status();
}

// half: the layers of `a` carry fp16 planes and scales (kHalf3); maxima: some layer of `a` leaves per-sample maxima
static int launch(const Args& a, hipStream_t stream, bool half = false) {
    const long n_blocks = (a.Ppad / 32 + 7) / 8;
    const long cap = max_workgroups();
    const unsigned G = (unsigned)(n_blocks < cap ? n_blocks : cap);
    bool maxima = false;
    for (int l = 0; l < a.n_layers; ++l) maxima = maxima || a.layer[l].amax_out;
    // a chain: what a layer stores is read back two blocks later -- default cache policy
    if (half) return a.n_layers > 1 ? launch_as<kHalf3 | kAmaxOut | kPlainStore>(a, G, stream) : launch_as<kHalf3 | kAmaxOut>(a, G, stream);
    if (maxima) return launch_as<kAmaxOut>(a, G, stream);
    return a.n_layers > 1 ? launch_as<kPlainStore>(a, G, stream) : launch_as<0>(a, G, stream);
}

// A chain goes out as ONE launch when every workgroup owns at least two 256-sample blocks (the pipeline then fetches
// the next layer's first block long after it was stored); otherwise layer by layer.
static int launch_chain(const Args& chain, hipStream_t stream, bool half = false) {
    const long n_blocks = (chain.Ppad / 32 + 7) / 8;
    if (chain.n_layers == 1 || n_blocks >= 2L * max_workgroups()) return launch(chain, stream, half);
    for (int l = 0; l < chain.n_layers; ++l) {
        Args one = chain;
        one.layer[0] = chain.layer[l];
        one.n_layers = 1;
        const int rc = launch(one, stream, half);
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
    L.amax_in = nullptr;
    L.amax_out = nullptr;
    L.w_inv_scale = nullptr;
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
    L.amax_in = nullptr;
    L.amax_out = nullptr;
    L.w_inv_scale = nullptr;
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
// `amax` (or nullptr): workspace of scnerf_layer_amax_floats -- with it the layers whose input comes with per-sample
// maxima run on THREE fp16 products (layer_split.h, kHalf3): layer 1 (its input comes from the fused stage) and the
// skip layer (a second operand with a range of its own) stay on six bf16 products and leave the maxima of what they
// store; 2-4 and 6-8 are fp16 chains.  The fp16 planes and the weight scales follow the bf16 planes in `planes`.
template <int PD>
int launch_network_chain_fwd(const short* planes, const float* wpacked, float* save, float* amax, long P, hipStream_t stream) {
    using V = Var<PD>;
    const long Ppad = padded_samples(P);
    unsigned* masks = reinterpret_cast<unsigned*>(save + (long)V::kSavePerSample * Ppad);
    const float* epts = save + (long)kSaveEpts * Ppad;
    const short* planes16 = planes + (long)total_slabs<PD>() * kSlabShorts;
    const float* w_inv = reinterpret_cast<const float*>(planes16 + (long)total_slabs<PD>() * kSlabShorts);
    auto layer = [&](int l, bool half) {
        Layer L = forward_layer<PD>(l, half ? planes16 : planes, wpacked + (l < 8 ? V::kFwdBias + 256 * l : V::kFwdBiasF),
                                    save + (long)(kSaveAct + 256 * (l - 1)) * Ppad, epts,
                                    save + (long)(l < 8 ? kSaveAct + 256 * l : kSaveFeat) * Ppad,
                                    l < 8 ? masks + (long)l * (Ppad / 32) * 256 : nullptr);
        if (amax) {
            if (half) { L.amax_in = amax + (long)(l - 2) * 2 * Ppad; L.w_inv_scale = w_inv + (l - 1); }
            if (l < 8) L.amax_out = amax + (long)(l - 1) * 2 * Ppad;
        }
        return L;
    };
    if (!amax) {
        Args a = chain_header(8, V::kEW, Ppad, 0, 0, 0);
        for (int l = 1; l <= 8; ++l) a.layer[l - 1] = layer(l, false);
        return launch_chain(a, stream);
    }
    const int groups[4][2] = {{1, 1}, {2, 4}, {5, 5}, {6, 8}};
    for (int g = 0; g < 4; ++g) {
        const bool half = g & 1;
        Args a = chain_header(groups[g][1] - groups[g][0] + 1, V::kEW, Ppad, 0, 0, 0);
        for (int l = groups[g][0]; l <= groups[g][1]; ++l) a.layer[l - groups[g][0]] = layer(l, half);
        const int rc = launch_chain(a, stream, half);
        if (rc) return rc;
    }
    return 0;
}

// entries 0 .. 7 of the data-gradient chain over the gradient workspace: d feature -> dZ_7 -> ... -> dZ_0
// (amax as above: entry 0 -- its input comes from the fused stage -- on six bf16 products, entries 1 .. 7 one fp16 chain)
template <int PD>
int launch_network_chain_bwd(const short* planes, const float* wpacked_bwd, const float* save, float* grads,
                             const float* d_raw, float* amax, long P, hipStream_t stream) {
    using V = Var<PD>;
    const long Ppad = padded_samples(P);
    const unsigned* masks = reinterpret_cast<const unsigned*>(save + (long)V::kSavePerSample * Ppad);
    const float* alpha = wpacked_bwd + V::kBwdAlphaW;
    const short* planes16 = planes + (long)total_slabs<PD>() * kSlabShorts;
    const float* w_inv = reinterpret_cast<const float*>(planes16 + (long)total_slabs<PD>() * kSlabShorts);
    auto layer = [&](int e, bool half) {
        Layer L = backward_layer<PD>(e, half ? planes16 : planes, alpha,
                                     grads + (long)(e == 0 ? kGradDfeat : kGradDz + (8 - e) * 256) * Ppad,
                                     grads + (long)(kGradDz + (7 - e) * 256) * Ppad,
                                     masks + (long)(7 - e) * (Ppad / 32) * 256, d_raw + 3);
        if (amax) {
            if (half) { L.amax_in = amax + (long)(e - 1) * 2 * Ppad; L.w_inv_scale = w_inv + (e == 0 ? 7 : 7 - e); }
            if (e < 7) L.amax_out = amax + (long)e * 2 * Ppad;
        }
        return L;
    };
    if (!amax) {
        Args a = chain_header(8, V::kEW, Ppad, 1, 4, P);
        for (int e = 0; e < 8; ++e) a.layer[e] = layer(e, false);
        return launch_chain(a, stream);
    }
    Args first = chain_header(1, V::kEW, Ppad, 1, 4, P);
    first.layer[0] = layer(0, false);
    int rc = launch_chain(first, stream);
    if (rc) return rc;
    Args a = chain_header(7, V::kEW, Ppad, 1, 4, P);
    for (int e = 1; e < 8; ++e) a.layer[e - 1] = layer(e, true);
    return launch_chain(a, stream, true);
}

template int launch_network_layer<3>(int, const short*, const float*, const float*, const float*, float*, unsigned*, long, hipStream_t);
template int launch_network_layer<4>(int, const short*, const float*, const float*, const float*, float*, unsigned*, long, hipStream_t);
template int launch_network_layer_bwd<3>(int, const short*, const float*, const float*, float*, const unsigned*, const float*, int, long, long, hipStream_t);
template int launch_network_layer_bwd<4>(int, const short*, const float*, const float*, float*, const unsigned*, const float*, int, long, long, hipStream_t);
template int launch_network_chain_fwd<3>(const short*, const float*, float*, float*, long, hipStream_t);
template int launch_network_chain_fwd<4>(const short*, const float*, float*, float*, long, hipStream_t);
template int launch_network_chain_bwd<3>(const short*, const float*, const float*, float*, const float*, float*, long, hipStream_t);
template int launch_network_chain_bwd<4>(const short*, const float*, const float*, float*, const float*, float*, long, hipStream_t);

}  // namespace lsp
}  // namespace scn

extern "C" int scnerf_layer_split_workgroups(int n) {
    if (n >= 1 && n <= 1024) scn::lsp::max_workgroups() = n;
    return scn::lsp::max_workgroups();
}

extern "C" long long scnerf_layer_amax_floats(long long n_samples) {
    return n_samples < 0 ? -1 : 7LL * 2 * padded_samples(n_samples);
}

extern "C" int scnerf_layer_split_chain_fwd(int pt_dims, const short* planes, const float* wpacked, float* save,
                                            float* amax, long long n_samples, void* stream) {
    SCN_RETURN_IF(!planes || !wpacked || !save || n_samples < 0 || (pt_dims != 3 && pt_dims != 4), SCN_EINVAL);
    if (n_samples == 0) return 0;
    return pt_dims == 3 ? scn::lsp::launch_network_chain_fwd<3>(planes, wpacked, save, amax, (long)n_samples, (hipStream_t)stream)
                        : scn::lsp::launch_network_chain_fwd<4>(planes, wpacked, save, amax, (long)n_samples, (hipStream_t)stream);
}

extern "C" int scnerf_layer_split_chain_bwd(int pt_dims, const short* planes, const float* wpacked_bwd, const float* save,
                                            float* grads, const float* d_raw, float* amax, long long n_samples, void* stream) {
    SCN_RETURN_IF(!planes || !wpacked_bwd || !save || !grads || !d_raw || n_samples < 0 || (pt_dims != 3 && pt_dims != 4), SCN_EINVAL);
    if (n_samples == 0) return 0;
    return pt_dims == 3 ? scn::lsp::launch_network_chain_bwd<3>(planes, wpacked_bwd, save, grads, d_raw, amax, (long)n_samples, (hipStream_t)stream)
                        : scn::lsp::launch_network_chain_bwd<4>(planes, wpacked_bwd, save, grads, d_raw, amax, (long)n_samples, (hipStream_t)stream);
}

// [bf16 planes][fp16 planes][8 x 1 / weight scale][8 x weight scale] (the last two as floats)
extern "C" long long scnerf_split_planes_shorts(int pt_dims) {
    if (pt_dims != 3 && pt_dims != 4) return -1;
    return 2LL * (pt_dims == 3 ? total_slabs<3>() : total_slabs<4>()) * scn::lsp::kSlabShorts + 32;
}

extern "C" int scnerf_pack_split_planes(int pt_dims, const float* flat_params, short* planes, void* stream) {
    SCN_RETURN_IF(!flat_params || !planes || (pt_dims != 3 && pt_dims != 4), SCN_EINVAL);
    const long slabs = pt_dims == 3 ? total_slabs<3>() : total_slabs<4>();
    const long n = slabs * 8 * 64 * 8;
    short* planes16 = planes + slabs * scn::lsp::kSlabShorts;
    float* tail = reinterpret_cast<float*>(planes16 + slabs * scn::lsp::kSlabShorts);
    hipStream_t st = (hipStream_t)stream;
    if (pt_dims == 3) {
        hipLaunchKernelGGL((pack_planes_kernel<3, false>), dim3(scn_ceil_div(n, 256)), dim3(256), 0, st, flat_params, planes, nullptr);
        hipLaunchKernelGGL(weight_scale_kernel<3>, dim3(8), dim3(256), 1024, st, flat_params, tail);
        hipLaunchKernelGGL((pack_planes_kernel<3, true>), dim3(scn_ceil_div(n, 256)), dim3(256), 0, st, flat_params, planes16, tail + 8);
    } else {
        hipLaunchKernelGGL((pack_planes_kernel<4, false>), dim3(scn_ceil_div(n, 256)), dim3(256), 0, st, flat_params, planes, nullptr);
        hipLaunchKernelGGL(weight_scale_kernel<4>, dim3(8), dim3(256), 1024, st, flat_params, tail);
        hipLaunchKernelGGL((pack_planes_kernel<4, true>), dim3(scn_ceil_div(n, 256)), dim3(256), 0, st, flat_params, planes16, tail + 8);
    }
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
