// randoms.hip -- the random numbers of one render_rays call in ONE launch.
//
// The reference draws four tensors per call with torch.rand / torch.randn (NeRF/render.py:252-257 stratified jitter t_rand,
// :425-429 the inverse-cdf variates u, :329-330 the density noise of both stages): on the device that is four generator
// launches plus the scaling passes of `randn * raw_noise_std`.  Here one kernel writes all of them, already scaled:
// Philox4x32-10 (Salmon et al., SC'11; the counter-based generator torch's own device generator is built on), key = the 64-bit
// seed, counter = (quad index within the stream, stream id, call counter lo, call counter hi) -- every value is a pure
// function of (seed, call, stream, element), so a run is reproducible from torch.manual_seed and independent of the launch
// geometry.  Uniforms: the top 24 bits x 2^-24, in [0, 1) as torch.rand's floats.  Normals: Box-Muller on two uniforms
// (the first shifted into (0, 1]).  The parity tests inject their draws through `_randoms` and never come here.
#include <hip/hip_runtime.h>

#include "launch.h"
#include "scnerf_hip.h"

namespace {

struct U4 { unsigned x, y, z, w; };

__host__ __device__ inline unsigned mulhi32(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

__host__ __device__ inline U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = mulhi32(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = mulhi32(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

__device__ inline float uniform01(unsigned x) { return (float)(x >> 8) * (1.f / 16777216.f); }            // [0, 1)
__device__ inline float uniform_open0(unsigned x) { return (float)((x >> 8) + 1u) * (1.f / 16777216.f); } // (0, 1]

struct Streams {
    float* out[4];          // t_rand, u, noise_c, noise_f (NULL: not wanted)
    long long n[4];         // elements
    long long quad0[5];     // first quad of each stream in the launch's quad numbering
    float std[4];           // 0: uniform in [0, 1); > 0: normal x std
};

__global__ __launch_bounds__(256) void render_randoms_kernel(Streams s, unsigned k0, unsigned k1, unsigned call_lo, unsigned call_hi) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= s.quad0[4]) return;
    int id = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) id += q >= s.quad0[i];
    const long long local = q - s.quad0[id];
    const U4 r = philox4x32_10(U4{(unsigned)local, (unsigned)(local >> 32) ^ ((unsigned)id << 28), call_lo, call_hi}, k0, k1);
    float v[4];
    if (s.std[id] > 0.f) {
        const float r0 = sqrtf(-2.f * logf(uniform_open0(r.x))), r1 = sqrtf(-2.f * logf(uniform_open0(r.z)));
        float s0, c0, s1, c1;
        sincosf(6.283185307179586f * uniform01(r.y), &s0, &c0);
        sincosf(6.283185307179586f * uniform01(r.w), &s1, &c1);
        v[0] = r0 * c0 * s.std[id]; v[1] = r0 * s0 * s.std[id]; v[2] = r1 * c1 * s.std[id]; v[3] = r1 * s1 * s.std[id];
    } else {
        v[0] = uniform01(r.x); v[1] = uniform01(r.y); v[2] = uniform01(r.z); v[3] = uniform01(r.w);
    }
    float* o = s.out[id] + 4 * local;
    const long long left = s.n[id] - 4 * local;
    if (left >= 4) {
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int i = 0; i < (int)left; ++i) o[i] = v[i];
    }
}

}  // namespace

extern "C" int scnerf_render_randoms(unsigned long long seed, unsigned long long call, float* t_rand, long long n_t_rand,
                                     float* u, long long n_u, float* noise_c, long long n_noise_c, float* noise_f,
                                     long long n_noise_f, float raw_noise_std, void* stream) {
    Streams s;
    float* outs[4] = {t_rand, u, noise_c, noise_f};
    const long long ns[4] = {n_t_rand, n_u, n_noise_c, n_noise_f};
    long long quads = 0;
    for (int i = 0; i < 4; ++i) {
        SCN_RETURN_IF(ns[i] < 0 || (outs[i] != nullptr && (reinterpret_cast<unsigned long long>(outs[i]) & 15ull)), SCN_EINVAL);
        s.out[i] = outs[i];
        s.n[i] = outs[i] ? ns[i] : 0;
        s.std[i] = i < 2 ? 0.f : raw_noise_std;
        s.quad0[i] = quads;
        quads += (s.n[i] + 3) / 4;
    }
    s.quad0[4] = quads;
    SCN_RETURN_IF(raw_noise_std < 0.f, SCN_EINVAL);
    if (quads == 0) return 0;
    hipLaunchKernelGGL(render_randoms_kernel, dim3(scn_ceil_div(quads, 256)), dim3(256), 0, (hipStream_t)stream, s,
                       (unsigned)seed, (unsigned)(seed >> 32), (unsigned)call, (unsigned)(call >> 32));
    return scn_launch_status();
}
