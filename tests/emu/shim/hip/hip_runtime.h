// TEST INFRASTRUCTURE -- a stand-in for <hip/hip_runtime.h> used ONLY when the kernel
// sources are compiled for the CPU SIMT interpreter (tests/emu).  Never on a product
// include path.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <algorithm>
#include "../simt_emu.h"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static_assert(false, "use scn::dynamic_lds<T>()");

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

#define threadIdx (simt::cur_thread_idx())
#define blockIdx (simt::cur_block_idx())
#define blockDim (simt::cur_block_dim())
#define gridDim (simt::cur_grid_dim())

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
enum hipFuncAttribute_emu { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...)                              \
    do {                                                                                       \
        dim3 g__ = (grid), b__ = (block);                                                      \
        simt::launch({g__.x, g__.y, g__.z}, {b__.x, b__.y, b__.z}, (size_t)(shmem),             \
                     [=]() { kern(__VA_ARGS__); });                                            \
    } while (0)

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; std::memcpy(&i, &f, 4); return i; }
using std::pow;
using std::sqrt;
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct float3 { float x, y, z; };
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return {x, y}; }
