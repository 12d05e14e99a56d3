// embed.hip -- stand-alone positional encoding (the reference's Embedder.embed, /root/reference
// NeRF/run_nerf_helpers.py:24-55):  x [n, d]  ->  [x, sin(x f_0), cos(x f_0), sin(x f_1), ...]  [n, d (1 + 2 F)]
// (column blocks of d in the order the reference concatenates them; `include_input` drops the first block).
// The render path never calls this -- the fused network kernels build the encoding in registers
// (mlp_common.h: pe_slots) -- it exists so that `embed_fn(x)` of the reference API works on its own, forward and
// backward.  One thread per (row, input component); x f is formed exactly as the reference does (one fp32
// multiply), sin / cos by the device's sincosf.
#include <scn_wave.h>

#include "launch.h"
#include "scnerf_hip.h"

namespace {

__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ x, long n, int d,
                                                        const float* __restrict__ freqs, int n_freqs, int include_input,
                                                        float* __restrict__ out) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * d) return;
    const long row = e / d;
    const int j = (int)(e - row * d);
    const int width = d * (include_input ? 1 : 0) + 2 * d * n_freqs;
    float* o = out + row * width;
    const float v = x[e];
    int col = 0;
    if (include_input) { o[j] = v; col = d; }
    for (int f = 0; f < n_freqs; ++f) {
        float s, c;
        scn::sincos(v * freqs[f], &s, &c);
        o[col + j] = s;
        o[col + d + j] = c;
        col += 2 * d;
    }
}

// g_x[row][j] = g[row][j] (if include_input) + sum_f f (g_sin cos(x f) - g_cos sin(x f))
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, long n,
                                                        int d, const float* __restrict__ freqs, int n_freqs,
                                                        int include_input, float* __restrict__ g_x) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * d) return;
    const long row = e / d;
    const int j = (int)(e - row * d);
    const int width = d * (include_input ? 1 : 0) + 2 * d * n_freqs;
    const float* gr = g + row * width;
    const float v = x[e];
    float acc = 0.f;
    int col = 0;
    if (include_input) { acc = gr[j]; col = d; }
    for (int f = 0; f < n_freqs; ++f) {
        float s, c;
        const float fr = freqs[f];
        scn::sincos(v * fr, &s, &c);
        acc += fr * (gr[col + j] * c - gr[col + d + j] * s);
        col += 2 * d;
    }
    g_x[e] = acc;
}

}  // namespace

extern "C" int scnerf_embed_fwd(const float* x, long long n, int d, const float* freqs, int n_freqs, int include_input,
                                float* out, void* stream) {
    SCN_RETURN_IF(!x || !out || (n_freqs > 0 && !freqs) || n < 0 || d < 1 || n_freqs < 0, SCN_EINVAL);
    if (n > 0)
        hipLaunchKernelGGL(embed_fwd_kernel, dim3(scn_ceil_div(n * d, 256)), dim3(256), 0, (hipStream_t)stream, x, (long)n, d,
                           freqs, n_freqs, include_input, out);
    return scn_launch_status();
}

extern "C" int scnerf_embed_bwd(const float* x, const float* g_out, long long n, int d, const float* freqs, int n_freqs,
                                int include_input, float* g_x, void* stream) {
    SCN_RETURN_IF(!x || !g_out || !g_x || (n_freqs > 0 && !freqs) || n < 0 || d < 1 || n_freqs < 0, SCN_EINVAL);
    if (n > 0)
        hipLaunchKernelGGL(embed_bwd_kernel, dim3(scn_ceil_div(n * d, 256)), dim3(256), 0, (hipStream_t)stream, x, g_out,
                           (long)n, d, freqs, n_freqs, include_input, g_x);
    return scn_launch_status();
}
