// Micro-benchmark: cost of independent VALU instructions issued between MFMAs by the same wave
// (one wave per SIMD): time per MFMA as a function of K VALU ops per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[8];
#pragma unroll
    for (int w = 0; w < 8; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[w][r] = (float)(w + r);
    float a = a0 + threadIdx.x, b = b0;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = a0 * i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                acc[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[w], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < K; ++q) v[q % 16] = fmaxf(v[q % 16] * 1.0001f, b0);      // 2 VALU each (mul, max)
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[w][r];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K>
void run(float* d) {
    const int iters = 1000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_wave = (double)iters * 64;
    printf("VALU per MFMA %2d: %.1f ns per MFMA per wave (%.1f cycles @2.4GHz)\n", 2 * K, ms * 1e6 / mfma_per_wave,
           ms * 1e6 / mfma_per_wave * 2.4);
}

int main() {
    float* d; (void)hipMalloc(&d, 256 * 256 * 4);
    run<0>(d); run<1>(d); run<2>(d); run<4>(d); run<6>(d); run<8>(d); run<12>(d);
    return 0;
}
