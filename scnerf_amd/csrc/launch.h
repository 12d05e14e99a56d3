// launch.h -- host-side helpers shared by the C-ABI entry points.
#pragma once
#include <hip/hip_runtime.h>

#define SCN_RETURN_IF(cond, code) \
    do {                          \
        if (cond) return (code);  \
    } while (0)

// argument errors are negative so they cannot be confused with a hipError_t
enum { SCN_EINVAL = -22, SCN_ENOSUP = -95 };

static inline int scn_launch_status() { return (int)hipGetLastError(); }

static inline unsigned scn_ceil_div(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// host API calls whose failure must surface as the entry point's status
#define SCN_HIP(call)                         \
    do {                                      \
        const hipError_t e__ = (call);        \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// Kernels that need more than 64 KB of dynamic LDS must opt in once per device (the attribute is a property
// of the function ON a device).  `mask` is a per-kernel static bit set indexed by the device ordinal.
#define SCN_LDS_OPT_IN(kernel_ptr, bytes)                                                              \
    do {                                                                                               \
        static unsigned long long mask__ = 0ull;                                                       \
        int dev__ = 0;                                                                                 \
        SCN_HIP(hipGetDevice(&dev__));                                                                 \
        if (!((mask__ >> (dev__ & 63)) & 1ull)) {                                                      \
            SCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_ptr),                     \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));    \
            mask__ |= 1ull << (dev__ & 63);                                                            \
        }                                                                                              \
    } while (0)
