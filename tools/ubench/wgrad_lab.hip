// Timing lab for the 256 x 256 weight-gradient GEMM (scnerf_amd/csrc/wgrad256.h): runs the product kernel
// and its ablation instantiations on synthetic tile-native operands and prints ms / TFLOP/s per GEMM.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iscnerf_amd/csrc/device -Iscnerf_amd/csrc \
//         -Iinclude tools/ubench/wgrad_lab.hip -o tools/ubench/wgrad_lab && tools/ubench/wgrad_lab [P] [jobs]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wgrad256.h"

using namespace scn::wg256;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)(i * 2654435761u) ^ seed;
        x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        p[i] = ((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f);
    }
}

__global__ void diff_kernel(const float* a, const float* b, size_t n, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.f;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(a[i] - b[i]));
    atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));
}

template <int FLAGS>
float run(const Args& a, int G, int iters) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad256_kernel<FLAGS>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w)
        hipLaunchKernelGGL((wgrad256_kernel<FLAGS>), dim3(G, a.n_jobs), dim3(kThreads), kLdsBytes, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it)
        hipLaunchKernelGGL((wgrad256_kernel<FLAGS>), dim3(G, a.n_jobs), dim3(kThreads), kLdsBytes, 0, a);
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
    chunk = (chunk + kMS - 1) / kMS * kMS;
    std::vector<float*> A(max_jobs), B(max_jobs);
    float *pw, *pw2, *pb, *dmax;
    const size_t slab = (size_t)G * kW * kW;
    for (int j = 0; j < max_jobs; ++j) {
        CK(hipMalloc(&A[j], (size_t)Ppad * kW * 4));
        CK(hipMalloc(&B[j], (size_t)Ppad * kW * 4));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, A[j], (size_t)Ppad * kW, 17u + j);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, B[j], (size_t)Ppad * kW, 1017u + j);
    }
    CK(hipMalloc(&pw, slab * 4 * max_jobs));
    CK(hipMalloc(&pw2, slab * 4));
    CK(hipMalloc(&pb, (size_t)G * kW * 4 * max_jobs));
    CK(hipMalloc(&dmax, 4));
    CK(hipDeviceSynchronize());
    auto make = [&](int n_jobs, float* w) {
        Args a;
        a.n_jobs = n_jobs; a.Ppad = Ppad; a.chunk = chunk;
        for (int j = 0; j < n_jobs; ++j) a.job[j] = Job{A[j], B[j], w + (n_jobs > 1 ? j * slab : 0), pb + (size_t)j * G * kW};
        return a;
    };
    const double flop = 2.0 * 256 * 256 * (double)P;
    auto report = [&](const char* name, float ms, int n_jobs) {
        printf("%-44s jobs=%d  %8.3f ms  %8.3f ms/GEMM  %6.1f TFLOP/s\n", name, n_jobs, ms, ms / n_jobs,
               flop * n_jobs / ms / 1e9);
        fflush(stdout);
    };
    printf("P = %ld (padded %ld), G = %d, chunk = %ld samples = %ld stages per workgroup\n", P, Ppad, G, chunk, chunk / kMS);
    const int it1 = 10, itn = 3;
    // correctness of the spread schedule on the hardware (LDS double buffering): same slab as the burst schedule
    {
        Args a0 = make(1, pw), a1 = make(1, pw2);
        run<0>(a0, G, 1);
        run<kSpread>(a1, G, 1);
        CK(hipMemset(dmax, 0, 4));
        hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, 0, pw, pw2, slab, dmax);
        float d; CK(hipMemcpy(&d, dmax, 4, hipMemcpyDeviceToHost));
        printf("max |burst - spread| over the partial slabs: %g\n", d);
    }
    report("burst (loads / commits in two bursts)", run<0>(make(1, pw), G, it1), 1);
    report("spread (product)", run<kSpread>(make(1, pw), G, it1), 1);
    report("spread, no bias sums", run<kSpread | kNoBias>(make(1, pw), G, it1), 1);
    report("spread, no barriers", run<kSpread | kNoBarrier>(make(1, pw), G, it1), 1);
    report("no loads / commits (LDS reads + MFMA + barrier)", run<kSpread | kNoLoad>(make(1, pw), G, it1), 1);
    report("no loads, no barriers (LDS reads + MFMA)", run<kSpread | kNoLoad | kNoBarrier>(make(1, pw), G, it1), 1);
    if (max_jobs > 1) {
        report("burst", run<0>(make(max_jobs, pw), G, itn), max_jobs);
        report("spread (product)", run<kSpread>(make(max_jobs, pw), G, itn), max_jobs);
        report("spread, no bias sums", run<kSpread | kNoBias>(make(max_jobs, pw), G, itn), max_jobs);
        report("no loads, no barriers", run<kSpread | kNoLoad | kNoBarrier>(make(max_jobs, pw), G, itn), max_jobs);
    }
    return 0;
}
