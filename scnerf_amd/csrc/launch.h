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
