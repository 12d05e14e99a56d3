// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 as a function of how many independent accumulators are
// interleaved (dependent-accumulate chains), one wave per SIMD like the MLP kernels.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_dep mfma_dep.hip && ./mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAYS>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[WAYS];
#pragma unroll
    for (int w = 0; w < WAYS; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[w][r] = (float)(w + r);
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 64 / WAYS; ++rep)
#pragma unroll
            for (int w = 0; w < WAYS; ++w) acc[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[w], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WAYS; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[w][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WAYS>
void run(float* d) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<WAYS>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<WAYS>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_wave = (double)iters * 64;
    const double flops = mfma_per_wave * 4 * blocks * 2.0 * 32 * 32 * 2;
    printf("ways %d: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per wave\n", WAYS, ms, flops / ms / 1e9, ms * 1e6 / mfma_per_wave);
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    run<1>(d); run<2>(d); run<4>(d); run<8>(d);
    return 0;
}
