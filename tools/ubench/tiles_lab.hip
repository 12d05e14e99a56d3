// Timing lab for the narrow weight-gradient shapes (scnerf_amd/csrc/wgrad_tiles.h): product kernel and its
// ablations on synthetic operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iscnerf_amd/csrc/device -Iscnerf_amd/csrc \
//         -Iinclude tools/ubench/tiles_lab.hip -o tools/ubench/tiles_lab && tools/ubench/tiles_lab [P]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "wgrad_tiles.h"

using namespace scn::wgt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)(i * 2654435761u) ^ seed;
        x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
        p[i] = ((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f);
    }
}

template <int NA, int NB, bool RM, int FLAGS>
void run(const char* name, const Args& a, int G, long P) {
    constexpr unsigned lds = lds_bytes<NA, NB, RM>();
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tiles_kernel<NA, NB, RM, FLAGS>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((wgrad_tiles_kernel<NA, NB, RM, FLAGS>), dim3(G), dim3(kThreads), lds, 0, a);
    CK(hipDeviceSynchronize());
    const int iters = 10;
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((wgrad_tiles_kernel<NA, NB, RM, FLAGS>), dim3(G), dim3(kThreads), lds, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double flop = 2.0 * NA * NB * (double)P, bytes = 4.0 * (NA + NB) * (double)P;
    printf("%3d x %3d  %-34s G=%4d  %7.3f ms  %6.1f TFLOP/s  %5.2f TB/s\n", NA, NB, name, G, ms, flop / ms / 1e9, bytes / ms / 1e9);
    fflush(stdout);
}

template <int NA, int NB, bool RM>
void shape(long P, float* A, float* B, float* pw, float* pb) {
    for (int G : {256, 512}) {
        Args a{A, B, pw, pb, P, (P + 127) / 128 * 128, 0};
        run<NA, NB, RM, 0>("product", a, G, P);
        run<NA, NB, RM, kNoBias>("no bias sums", a, G, P);
        run<NA, NB, RM, kNoBarrier>("no barriers", a, G, P);
        run<NA, NB, RM, kNoLoad>("no loads / commits", a, G, P);
        run<NA, NB, RM, kNoLoad | kNoBarrier>("no loads, no barriers", a, G, P);
    }
}

int main(int argc, char** argv) {
    const long P = argc > 1 ? atol(argv[1]) : 786432;
    float *A, *B, *pw, *pb;
    CK(hipMalloc(&A, (size_t)P * 256 * 4 + 4096));
    CK(hipMalloc(&B, (size_t)P * 256 * 4 + 4096));
    CK(hipMalloc(&pw, (size_t)512 * 256 * 256 * 4));
    CK(hipMalloc(&pb, (size_t)512 * 256 * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, A, (size_t)P * 256, 17u);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, B, (size_t)P * 256, 1017u);
    CK(hipDeviceSynchronize());
    printf("P = %ld\n", P);
    shape<256, 64, true>(P, A, B, pw, pb);
    shape<128, 256, false>(P, A, B, pw, pb);
    return 0;
}
