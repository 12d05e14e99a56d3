// Lab for the bf16-split 256 -> 256 layer GEMM (scnerf_amd/csrc/layer_split.h): accuracy against fp64 on sampled
// outputs, then ms per layer at P samples.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iscnerf_amd/csrc/device -Iscnerf_amd/csrc \
//         tools/ubench/layer_split_lab.hip -o tools/ubench/layer_split_lab && tools/ubench/layer_split_lab [P]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "layer_split.h"

using namespace scn::lsp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, int relu, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)(i * 2654435761u) ^ seed;
        x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        unsigned y = x * 0x9e3779b9u + 12345u;
        y ^= y >> 13; y *= 0x85ebca6bu; y ^= y >> 16;
        float v = ((int)(x & 0xffffff) - 8388608) * (1.0f / 8388608.0f) * scale;
        v *= exp2f((float)((int)(y & 3) - 2));
        if (relu && (y & 8)) v = 0.f;
        p[i] = v;
    }
}

__device__ inline size_t tile_native(long p, int f) {
    return (size_t)(p >> 5) * 8192 + (size_t)(((f >> 3) * 64) + (p & 31) + 32 * ((f >> 2) & 1)) * 4 + (f & 3);
}

// W [256][256] fp32 row-major -> planes in fragment order [slab][plane][T][lane][8]
__global__ void pack_planes_kernel(const float* W, short* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // (s, T, lane, e)
    if (idx >= 16 * 8 * 64 * 8) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, T = (idx >> 9) & 7, s = idx >> 12;
    const int n = 32 * T + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
    const float x = W[n * 256 + k];
    const unsigned u = __float_as_uint(x);
    const float d1 = x - __uint_as_float(u & 0xffff0000u);
    const unsigned u1 = __float_as_uint(d1);
    const float d2 = d1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned u2 = __float_as_uint(d2);
    const unsigned pl[3] = {u >> 16, u1 >> 16, u2 >> 16};
    for (int q = 0; q < 3; ++q) out[(((size_t)(s * 3 + q) * 8 + T) * 64 + lane) * 8 + e] = (short)pl[q];
}

// the fp16 prototype's planes: h = f16(w * scale), l = f16(w * scale - h) (the third plane stays empty)
__global__ void pack_planes_f16_kernel(const float* W, short* out, float scale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // (s, T, lane, e)
    if (idx >= 16 * 8 * 64 * 8) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, T = (idx >> 9) & 7, s = idx >> 12;
    const int n = 32 * T + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
    const float x = W[n * 256 + k] * scale;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    const short pl[3] = {__builtin_bit_cast(short, h), __builtin_bit_cast(short, l), 0};
    for (int q = 0; q < 3; ++q) out[(((size_t)(s * 3 + q) * 8 + T) * 64 + lane) * 8 + e] = pl[q];
}

__global__ void bias_table_kernel(const float* b, float* table) {
    const int i = threadIdx.x;      // 256 entries: ((4 t + q) * 2 + h) * 4 + j  <->  feature 32 t + 8 q + 4 h + j
    const int j = i & 3, h = (i >> 2) & 1, q = (i >> 3) & 3, t = i >> 5;
    table[i] = b[32 * t + 8 * q + 4 * h + j];
}

// sampled check: one thread per (sample p from a strided set, feature n)
__global__ void check_kernel(const float* X, const float* W, const float* b, const float* Z, long P, int relu, long stride,
                             double* max_err, double* max_scaled) {
    const long p = ((long)blockIdx.x * stride) % P;
    const int n = threadIdx.x;
    double s = b[n], sa = fabs((double)b[n]);
    for (int k = 0; k < 256; ++k) {
        const double x = (double)W[n * 256 + k] * (double)X[tile_native(p, k)];
        s += x; sa += fabs(x);
    }
    if (relu && s < 0) s = 0;
    const double got = Z[tile_native(p, n)];
    const double e = fabs(got - s);
    atomicMax(reinterpret_cast<unsigned long long*>(max_err), __double_as_longlong(e));
    atomicMax(reinterpret_cast<unsigned long long*>(max_scaled), __double_as_longlong(e / sa));
}

template <typename Kern>
float time_kernel(Kern kern, const Args& a, int G, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(G), dim3(256), kLdsBytes, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kern, dim3(G), dim3(256), kLdsBytes, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const long P = argc > 1 ? atol(argv[1]) : 786432;
    const int G = argc > 2 ? atoi(argv[2]) : 256;
    const long Ppad = (P + 127) / 128 * 128;
    float *X, *Z, *W, *b, *table;
    short* Wp;
    unsigned* mask;
    double* errs;
    CK(hipMalloc(&X, (size_t)Ppad * 256 * 4)); CK(hipMalloc(&Z, (size_t)Ppad * 256 * 4));
    CK(hipMalloc(&W, 65536 * 4)); CK(hipMalloc(&b, 1024)); CK(hipMalloc(&table, 1024));
    CK(hipMalloc(&Wp, (size_t)16 * kSlabShorts * 2)); CK(hipMalloc(&mask, (size_t)Ppad / 32 * 256 * 4));
    CK(hipMalloc(&errs, 16));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, X, (size_t)Ppad * 256, 17u, 1, 1.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, W, (size_t)65536, 99u, 0, 0.1f);
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(256), 0, 0, b, (size_t)256, 7u, 0, 0.5f);
    hipLaunchKernelGGL(pack_planes_kernel, dim3(16 * 8 * 64 * 8 / 256), dim3(256), 0, 0, W, Wp);
    hipLaunchKernelGGL(bias_table_kernel, dim3(1), dim3(256), 0, 0, b, table);
    CK(hipMemset(Z, 0xff, (size_t)Ppad * 256 * 4));
    CK(hipDeviceSynchronize());
    Args a;
    a.n_layers = 1; a.clock_probe = nullptr; a.x_scale = a.out_scale = 1.f; a.group = 1 << 20; a.x2_ld = 64; a.Ppad = Ppad; a.mode = 0; a.vec_stride = 0; a.n_vec = 0;
    a.layer[0] = Layer{X, X, 16, 1, Wp, table, Z, mask, nullptr, nullptr};
    // the same layer eight times in ONE launch, every layer into a buffer of its own (what a training pass does)
    float* Zs[8];
    for (int l = 0; l < 8; ++l) CK(hipMalloc(&Zs[l], (size_t)Ppad * 256 * 4));
    Args chain = a;
    chain.n_layers = 8;
    for (int l = 0; l < 8; ++l) {
        const float* in = l ? Zs[l - 1] : X;
        chain.layer[l] = Layer{in, in, 16, 1, Wp, table, Zs[l], mask, nullptr, nullptr};
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kNoEpilogue>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kNoCut>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kNoCut | kNoEpilogue>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    hipLaunchKernelGGL((layer_split_kernel<0>), dim3(G), dim3(256), kLdsBytes, 0, a);
    CK(hipDeviceSynchronize()); CK(hipGetLastError());
    CK(hipMemset(errs, 0, 16));
    hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, X, W, b, Z, P, 1, 7919L, errs, errs + 1);
    double h[2];
    CK(hipMemcpy(h, errs, 16, hipMemcpyDeviceToHost));
    printf("P = %ld, G = %d: 4096 sampled rows x 256 features vs fp64: max |err| = %.3e, max |err| / (|b| + sum |w x|) = %.3e\n", P, G, h[0], h[1]);
    // clocks
    for (int w = 0; w < 4; ++w) time_kernel(layer_split_kernel<0>, a, G, 20);
    const double flop = 2.0 * 256 * 256 * (double)P;
    auto report = [&](const char* name, float ms) {
        printf("%-44s %8.3f ms  %6.1f TFLOP/s  %5.2f TB/s (X read + Z write)\n", name, ms, flop / ms / 1e9, 2.0 * P * 1024 / ms / 1e9);
        fflush(stdout);
    };
    report("layer split x6", time_kernel(layer_split_kernel<0>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kPlainStore>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    for (int group : {1 << 20, 2, 3, 4, 6}) {
        chain.group = group;
        const float ms8 = time_kernel(layer_split_kernel<0>, chain, G, 8);
        const float ms8p = time_kernel(layer_split_kernel<kPlainStore>, chain, G, 8);
        printf("chained, groups of %-7d                     %8.3f ms per layer (8 layers in one launch: %.3f ms); plain stores %.3f\n",
               group, ms8 / 8, ms8, ms8p / 8);
        fflush(stdout);
    }
    {   // the chain's last layer against fp64 from its input (groups of 2)
        chain.group = 2;
        hipLaunchKernelGGL((layer_split_kernel<0>), dim3(G), dim3(256), kLdsBytes, 0, chain);
        CK(hipMemset(errs, 0, 16));
        hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, Zs[6], W, b, Zs[7], P, 1, 7919L, errs, errs + 1);
        CK(hipMemcpy(h, errs, 16, hipMemcpyDeviceToHost));
        printf("chain, groups of 2, layer 8 from layer 7 vs fp64: max |err| = %.3e, scaled %.3e\n", h[0], h[1]);
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kPlainStore>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, plain stores", time_kernel(layer_split_kernel<kPlainStore>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kNoZStore>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, epilogue without Z stores", time_kernel(layer_split_kernel<kNoZStore>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kNoCutMath | kNoEpilogue>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, X loaded but not cut, no epilogue", time_kernel(layer_split_kernel<kNoCutMath | kNoEpilogue>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kNoBarrier>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, no barriers (racy)", time_kernel(layer_split_kernel<kNoBarrier>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kSameX | kNoEpilogue>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, X always block 0 (L2 hits), no epilogue", time_kernel(layer_split_kernel<kSameX | kNoEpilogue>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kNoWCopy | kNoEpilogue>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, no W copy, no epilogue", time_kernel(layer_split_kernel<kNoWCopy | kNoEpilogue>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kNoWCopy | kSameX | kNoEpilogue>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, no W copy, X tile 0, no epilogue", time_kernel(layer_split_kernel<kNoWCopy | kSameX | kNoEpilogue>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kNoWCopy | kNoCut | kNoEpilogue>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, MFMA only (no W copy, no X)", time_kernel(layer_split_kernel<kNoWCopy | kNoCut | kNoEpilogue>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kNoDupX | kNoEpilogue>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, no two waves load the same X, no epilogue", time_kernel(layer_split_kernel<kNoDupX | kNoEpilogue>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kNoDupX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, no two waves load the same X", time_kernel(layer_split_kernel<kNoDupX>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kRotate | kNoEpilogue>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, K loop rotated per block, no epilogue", time_kernel(layer_split_kernel<kRotate | kNoEpilogue>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kRotate>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, K loop rotated per block", time_kernel(layer_split_kernel<kRotate>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kXfromW | kNoEpilogue>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, X loads fetch W pieces, no epilogue", time_kernel(layer_split_kernel<kXfromW | kNoEpilogue>, a, G, 40));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kRandomX | kNoCut | kNoEpilogue>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    report("layer split x6, MFMA + W stream, random static X planes", time_kernel(layer_split_kernel<kRandomX | kNoCut | kNoEpilogue>, a, G, 40));
    {   // the shader clock under each variant: shader cycles / 100 MHz reference ticks, workgroup 0 and the mean
        unsigned long long* probe;
        CK(hipMalloc(&probe, 16 * 256));
        Args ap = a;
        ap.clock_probe = probe;
        auto clock_of = [&](auto kern, const char* name) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
            const float ms = time_kernel(kern, ap, G, 40);
            unsigned long long h2[512];
            CK(hipMemcpy(h2, probe, 16 * G, hipMemcpyDeviceToHost));
            double cyc = 0, ref = 0;
            for (int i = 0; i < G; ++i) { cyc += (double)h2[2 * i]; ref += (double)h2[2 * i + 1]; }
            printf("%-60s %8.3f ms   shader clock %.3f GHz (cycles per workgroup %.0f)\n", name, ms, cyc / ref * 0.1, cyc / G);
            fflush(stdout);
        };
        clock_of(layer_split_kernel<kClockProbe>, "clock: full kernel");
        clock_of(layer_split_kernel<kClockProbe | kNoEpilogue>, "clock: no epilogue");
        clock_of(layer_split_kernel<kClockProbe | kNoCut | kNoEpilogue>, "clock: MFMA + W stream, garbage X planes");
        clock_of(layer_split_kernel<kClockProbe | kRandomX | kNoCut | kNoEpilogue>, "clock: MFMA + W stream, random static X planes");
    }
    {   // PROTOTYPE: two fp16 planes per operand, three products (see kHalf3); same data, same check against fp64
        short* Wp16;
        CK(hipMalloc(&Wp16, (size_t)16 * kSlabShorts * 2));
        const float sx = 4096.f, sw = 32768.f;
        hipLaunchKernelGGL(pack_planes_f16_kernel, dim3(16 * 8 * 64 * 8 / 256), dim3(256), 0, 0, W, Wp16, sw);
        Args h3 = a;
        h3.layer[0].W = Wp16;
        h3.x_scale = sx;
        h3.out_scale = 1.f / (sx * sw);
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kHalf3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
        CK(hipMemset(Z, 0xff, (size_t)Ppad * 256 * 4));
        hipLaunchKernelGGL((layer_split_kernel<kHalf3>), dim3(G), dim3(256), kLdsBytes, 0, h3);
        CK(hipDeviceSynchronize()); CK(hipGetLastError());
        CK(hipMemset(errs, 0, 16));
        hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, X, W, b, Z, P, 1, 7919L, errs, errs + 1);
        CK(hipMemcpy(h, errs, 16, hipMemcpyDeviceToHost));
        printf("fp16 x 3 products: 4096 sampled rows x 256 features vs fp64: max |err| = %.3e, max |err| / (|b| + sum |w x|) = %.3e\n", h[0], h[1]);
        report("fp16 x 3 products, one layer per launch", time_kernel(layer_split_kernel<kHalf3>, h3, G, 40));
        Args c3 = chain;
        c3.group = 2; c3.x_scale = sx; c3.out_scale = h3.out_scale;
        for (int l = 0; l < 8; ++l) c3.layer[l].W = Wp16;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(layer_split_kernel<kHalf3 | kPlainStore>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
        const float ms8 = time_kernel(layer_split_kernel<kHalf3 | kPlainStore>, c3, G, 8);
        printf("fp16 x 3 products, chained (groups of 2, plain stores) %8.3f ms per layer (8 layers in one launch: %.3f ms)\n", ms8 / 8, ms8);
        unsigned long long* probe;
        CK(hipMalloc(&probe, 16 * 256));
        h3.clock_probe = probe;
        auto clock_of = [&](auto kern, const char* name) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
            const float ms = time_kernel(kern, h3, G, 40);
            unsigned long long h2[512];
            CK(hipMemcpy(h2, probe, 16 * G, hipMemcpyDeviceToHost));
            double cyc = 0, ref = 0;
            for (int i = 0; i < G; ++i) { cyc += (double)h2[2 * i]; ref += (double)h2[2 * i + 1]; }
            printf("%-60s %8.3f ms   shader clock %.3f GHz (cycles per workgroup %.0f)\n", name, ms, cyc / ref * 0.1, cyc / G);
        };
        clock_of(layer_split_kernel<kHalf3 | kClockProbe>, "clock: fp16 x 3, full kernel");
        clock_of(layer_split_kernel<kHalf3 | kClockProbe | kNoEpilogue>, "clock: fp16 x 3, no epilogue");
        fflush(stdout);
    }
    report("layer split x6, no epilogue", time_kernel(layer_split_kernel<kNoEpilogue>, a, G, 40));
    report("layer split x6, no X loads / cuts", time_kernel(layer_split_kernel<kNoCut>, a, G, 40));
    report("layer split x6, MFMA + W stream only", time_kernel(layer_split_kernel<kNoCut | kNoEpilogue>, a, G, 40));
    return 0;
}
