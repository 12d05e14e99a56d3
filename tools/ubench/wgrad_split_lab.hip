// Lab for the bf16-split 256 x 256 weight-gradient GEMM (scnerf_amd/csrc/wgrad256_split.h) next to the exact-fp32
// MFMA kernel (wgrad256.h): accuracy of both against an fp64 reference on the same tile-native operands, then
// ms / TFLOP/s per GEMM.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iscnerf_amd/csrc/device -Iscnerf_amd/csrc \
//         tools/ubench/wgrad_split_lab.hip -o tools/ubench/wgrad_split_lab && tools/ubench/wgrad_split_lab [P] [jobs]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wgrad256.h"
#include "wgrad256_split.h"

namespace f32k = scn::wg256;
namespace spl = scn::wg256s;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// values with a wide spread of magnitudes and full 24-bit significands; `relu`: half of them zero (activations)
__global__ void fill_kernel(float* p, size_t n, unsigned seed, int relu) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)(i * 2654435761u) ^ seed;
        x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        unsigned y = x * 0x9e3779b9u + 12345u;
        y ^= y >> 13; y *= 0x85ebca6bu; y ^= y >> 16;
        float v = ((int)(x & 0xffffff) - 8388608) * (1.0f / 8388608.0f);          // 24 random bits in [-1, 1)
        v *= exp2f((float)((int)(y & 7) - 4));                                     // x 2^-4 .. 2^3
        if (relu && (y & 8)) v = 0.f;
        p[i] = v;
    }
}

__device__ inline size_t tile_native(long p, int f) {
    return (size_t)(p >> 5) * 8192 + (size_t)(((f >> 3) * 64) + (p & 31) + 32 * ((f >> 2) & 1)) * 4 + (f & 3);
}

// fp64 reference over the first `P` samples: one thread per output element
__global__ void ref_kernel(const float* A, const float* B, long P, double* dW, double* dWabs) {
    const int n = blockIdx.x, k = threadIdx.x;
    double s = 0, sa = 0;
    for (long p = 0; p < P; ++p) {
        const double x = (double)A[tile_native(p, n)] * (double)B[tile_native(p, k)];
        s += x; sa += fabs(x);
    }
    dW[n * 256 + k] = s;
    dWabs[n * 256 + k] = sa;
}

__global__ void reduce_kernel(const float* part, int G, double* out) {         // out[e] = sum_g part[g][e] in fp64
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    for (int g = 0; g < G; ++g) s += (double)part[(size_t)g * 65536 + e];
    out[e] = s;
}

template <typename Kern, typename Args>
float time_kernel(Kern kern, const Args& a, int G, unsigned lds, int iters) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(G, a.n_jobs), dim3(256), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kern, dim3(G, a.n_jobs), dim3(256), lds, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const long P = argc > 1 ? atol(argv[1]) : 786432;
    const int max_jobs = argc > 2 ? atoi(argv[2]) : 8;
    const int G = 256;
    const long Ppad = (P + 127) / 128 * 128;
    long chunk = (Ppad + G - 1) / G;
    chunk = (chunk + 31) / 32 * 32;
    std::vector<float*> A(max_jobs), B(max_jobs);
    const size_t slab = (size_t)G * 65536;
    float *pw, *pb;
    for (int j = 0; j < max_jobs; ++j) {
        CK(hipMalloc(&A[j], (size_t)Ppad * 256 * 4));
        CK(hipMalloc(&B[j], (size_t)Ppad * 256 * 4));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, A[j], (size_t)Ppad * 256, 17u + j, 0);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, B[j], (size_t)Ppad * 256, 1017u + j, 1);
    }
    CK(hipMalloc(&pw, slab * 4 * max_jobs));
    CK(hipMalloc(&pb, (size_t)G * 256 * 4 * max_jobs));
    CK(hipDeviceSynchronize());

    auto make_f32 = [&](int n_jobs) {
        f32k::Args a;
        a.n_jobs = n_jobs; a.Ppad = Ppad; a.chunk = chunk;
        for (int j = 0; j < n_jobs; ++j) a.job[j] = f32k::Job{A[j], B[j], pw + j * slab, pb + (size_t)j * G * 256};
        return a;
    };
    auto make_spl = [&](int n_jobs) {
        spl::Args a;
        a.n_jobs = n_jobs; a.Ppad = Ppad; a.chunk = chunk;
        for (int j = 0; j < n_jobs; ++j) a.job[j] = spl::Job{A[j], B[j], pw + j * slab, pb + (size_t)j * G * 256};
        return a;
    };

    // ---- accuracy on a problem the fp64 reference finishes quickly: Pa samples in 256 chunks
    {
        const long Pa = 32768;
        double *ref, *refabs, *got;
        CK(hipMalloc(&ref, 65536 * 8)); CK(hipMalloc(&refabs, 65536 * 8)); CK(hipMalloc(&got, 65536 * 8));
        hipLaunchKernelGGL(ref_kernel, dim3(256), dim3(256), 0, 0, A[0], B[0], Pa, ref, refabs);
        std::vector<double> r(65536), ra(65536), g(65536);
        CK(hipMemcpy(r.data(), ref, 65536 * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ra.data(), refabs, 65536 * 8, hipMemcpyDeviceToHost));
        std::vector<float> bias(256 * G);
        auto check = [&](const char* name) {
            hipLaunchKernelGGL(reduce_kernel, dim3(256), dim3(256), 0, 0, pw, G, got);
            CK(hipMemcpy(g.data(), got, 65536 * 8, hipMemcpyDeviceToHost));
            double worst = 0, rms = 0, worst_abs = 0;
            for (int e = 0; e < 65536; ++e) {
                const double d = fabs(g[e] - r[e]) / ra[e];          // relative to sum |a b|: the natural scale
                worst = fmax(worst, d); rms += d * d; worst_abs = fmax(worst_abs, fabs(g[e] - r[e]));
            }
            printf("%-40s max |err| / sum|ab| = %.3e   rms = %.3e   max |err| = %.3e\n", name, worst, sqrt(rms / 65536), worst_abs);
        };
        f32k::Args af = make_f32(1); af.Ppad = Pa; af.chunk = Pa / G;
        spl::Args as = make_spl(1); as.Ppad = Pa; as.chunk = Pa / G;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(f32k::wgrad256_kernel<f32k::kSpread>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)f32k::kLdsBytes));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spl::wgrad256_split_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)spl::kLdsBytes));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spl::wgrad256_split_kernel<spl::kNine>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)spl::kLdsBytes));
        printf("accuracy vs fp64, %ld samples in %d chunks of %ld (partials summed in fp64)\n", Pa, G, Pa / G);
        hipLaunchKernelGGL((f32k::wgrad256_kernel<f32k::kSpread>), dim3(G, 1), dim3(256), f32k::kLdsBytes, 0, af);
        CK(hipDeviceSynchronize()); check("exact-fp32 MFMA (32x32x2)");
        std::vector<float> b0(256 * G), b1(256 * G);
        CK(hipMemcpy(b0.data(), pb, b0.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(pw, 0, slab * 4));
        hipLaunchKernelGGL((spl::wgrad256_split_kernel<0>), dim3(G, 1), dim3(256), spl::kLdsBytes, 0, as);
        CK(hipDeviceSynchronize()); CK(hipGetLastError()); check("bf16 split, 6 products");
        CK(hipMemcpy(b1.data(), pb, b1.size() * 4, hipMemcpyDeviceToHost));
        double bd = 0, bs = 0;
        for (int n = 0; n < 256; ++n) {
            double s0 = 0, s1 = 0;
            for (int g2 = 0; g2 < G; ++g2) { s0 += b0[g2 * 256 + n]; s1 += b1[g2 * 256 + n]; }
            bd = fmax(bd, fabs(s0 - s1)); bs = fmax(bs, fabs(s0));
        }
        printf("bias sums: max |fp32 kernel - split kernel| = %.3e (scale %.3e)\n", bd, bs);
        CK(hipMemset(pw, 0, slab * 4));
        hipLaunchKernelGGL((spl::wgrad256_split_kernel<spl::kNine>), dim3(G, 1), dim3(256), spl::kLdsBytes, 0, as);
        CK(hipDeviceSynchronize()); check("bf16 split, 9 products");
    }

    const double flop = 2.0 * 256 * 256 * (double)P;
    auto report = [&](const char* name, float ms, int n_jobs) {
        printf("%-48s jobs=%d  %8.3f ms  %8.3f ms/GEMM  %6.1f TFLOP/s  %6.2f TB/s\n", name, n_jobs, ms, ms / n_jobs,
               flop * n_jobs / ms / 1e9, 2.0 * P * 1024 * n_jobs / ms / 1e9);
        fflush(stdout);
    };
    printf("P = %ld (padded %ld), G = %d, chunk = %ld samples\n", P, Ppad, G, chunk);
    const int it1 = 40, itn = 10;
    // clocks: two seconds of the fp32 kernel before anything is timed
    for (int w = 0; w < 6; ++w) time_kernel(f32k::wgrad256_kernel<f32k::kSpread>, make_f32(max_jobs), G, f32k::kLdsBytes, 40 / max_jobs + 1);
    report("exact-fp32 MFMA (product)", time_kernel(f32k::wgrad256_kernel<f32k::kSpread>, make_f32(1), G, f32k::kLdsBytes, it1), 1);
    report("bf16 split x6", time_kernel(spl::wgrad256_split_kernel<0>, make_spl(1), G, spl::kLdsBytes, it1), 1);
    report("bf16 split x6, no barriers", time_kernel(spl::wgrad256_split_kernel<spl::kNoBarrier>, make_spl(1), G, spl::kLdsBytes, it1), 1);
    report("bf16 split x6, no loads / cuts / commits", time_kernel(spl::wgrad256_split_kernel<spl::kNoLoad>, make_spl(1), G, spl::kLdsBytes, it1), 1);
    report("bf16 split x6, LDS reads + MFMA only", time_kernel(spl::wgrad256_split_kernel<spl::kNoLoad | spl::kNoBarrier>, make_spl(1), G, spl::kLdsBytes, it1), 1);
    report("bf16 split x9", time_kernel(spl::wgrad256_split_kernel<spl::kNine>, make_spl(1), G, spl::kLdsBytes, it1), 1);
    if (max_jobs > 1) {
        report("exact-fp32 MFMA (product)", time_kernel(f32k::wgrad256_kernel<f32k::kSpread>, make_f32(max_jobs), G, f32k::kLdsBytes, itn), max_jobs);
        report("bf16 split x6", time_kernel(spl::wgrad256_split_kernel<0>, make_spl(max_jobs), G, spl::kLdsBytes, itn), max_jobs);
    }
    return 0;
}
