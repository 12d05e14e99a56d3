// guest_write_lab.hip -- stand-alone reproducer for the "guest wave loses register writes" finding (profiles/r05_two_process_probe.txt):
// NO library code.  HOST kernel: a v_mfma_f32_32x32x16_f16 chain with a VALU epilogue and a global -> LDS operand stream, sized by
// an empty asm's clobbers to a chosen register allocation, looping on stream A.  GUEST kernel: a 64-register VALU + load kernel
// launched back to back on stream B of the SAME process (default) or in a process of its own (roles `host` / `guest`, started side
// by side by guest_write_lab_run.sh: the setting the finding was made in); every launch's output is compared bit for bit with the
// output of a solo launch.  One line per host mode:  mode, host registers, guest launches, launches that deviated, lanes (mod 64).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/guest_write_lab.hip -o tools/ubench/guest_write_lab
//   tools/ubench/guest_write_lab [seconds per mode = 6]  |  guest_write_lab host MODE SECONDS  |  guest_write_lab guest SECONDS LABEL
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// MODE 0: 424 registers (v255 + a167 claimed), one wave per SIMD   -- the shape of the product kernels before the claim
// MODE 1: 512 registers (v255 + a255)                              -- after claim_whole_register_file()
// MODE 2: 248 registers, launch bound two workgroups per CU, grid of ONE per CU -- the 256 x 64 narrow GEMM in a tail (256 free)
// MODE 3: as 0 without the epilogue and the stream (MFMAs only)
template <int MODE>
__global__ __launch_bounds__(256, MODE == 2 ? 2 : 1) void host_kernel(const float* __restrict__ w, float* __restrict__ out, int iters) {
    if (MODE == 0 || MODE == 3) asm volatile("" ::: "v255", "a167");
    if (MODE == 1) asm volatile("" ::: "v255", "a255");
    if (MODE == 2) asm volatile("" ::: "v247");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b[4];
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); for (int i = 0; i < 4; ++i) b[i][e] = (_Float16)(0.003f * (lane ^ (e + i))); }
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 3) {                                     // operand stream: 16 bytes per thread global -> LDS ring (32 KB)
            const f4 v = *reinterpret_cast<const f4*>(w + ((it & 63) * 256 + threadIdx.x) * 4);
            *reinterpret_cast<f4*>(lds + ((it & 7) * 256 + threadIdx.x) * 16) = v;
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // (the product's shape: every MFMA group's A fragment is a 16-byte LDS read issued between the MFMAs)
            if (MODE != 3) a = *reinterpret_cast<const h8*>(lds + (((it + k) & 31) * 64 + lane) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[i], acc[i], 0, 0, 0);
        }
        if (MODE != 3) {                                     // epilogue: ReLU, maximum, re-cut into the next B operand
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float z = fmaxf(acc[i][e] * 1e-3f + 0.125f, 0.f);
                    keep = fmaxf(keep, z);
                    b[i][e] = (_Float16)(z - (float)(_Float16)z + 0.003f * (lane ^ e));
                    acc[i][e] = 0.f;
                }
        }
    }
    float s = keep;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the guest: 48 live values per lane, rounds of (three-word load, dependent FMAs into every value), 64 registers or fewer
__global__ __launch_bounds__(256) void guest_kernel(const float* __restrict__ grid, float* __restrict__ out, int rounds) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float v[48];
    for (int k = 0; k < 48; ++k) v[k] = 0.001f * (float)((t + 7 * k) & 1023);
    for (int r = 0; r < rounds; ++r) {
        const float* g = grid + (((t * 131 + r * 977) & 16383) * 3);
        const float gx = g[0], gy = g[1], gz = g[2];
#pragma unroll
        for (int k = 0; k < 48; k += 3) {
            v[k] = __builtin_fmaf(v[k], 0.75f, gx);
            v[k + 1] = __builtin_fmaf(v[k + 1], 0.5f, gy * v[k]);
            v[k + 2] = __builtin_fmaf(v[k + 2], 0.25f, gz - v[k + 1]);
        }
    }
#pragma unroll
    for (int k = 0; k < 48; ++k) out[(long)k * gridDim.x * 256 + t] = v[k];
}

template <int MODE>
static int run_mode(const char* label, double seconds, const float* w, float* hout, const float* grid, float* gout, const std::vector<float>& solo,
                    hipStream_t sa, hipStream_t sb, int guest_blocks, bool with_host = true, bool with_guest = true) {
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(host_kernel<MODE>)));
    const size_t n = solo.size();
    std::vector<float> got(n);
    long launches = 0, bad = 0, lanes[64] = {0};
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int q = 0; q < 4 && with_host; ++q) hipLaunchKernelGGL(host_kernel<MODE>, dim3(256), dim3(256), 32768, sa, w, hout, 600);
        for (int q = 0; q < 8 && with_guest; ++q) {
            hipLaunchKernelGGL(guest_kernel, dim3(guest_blocks), dim3(256), 0, sb, grid, gout, 64);
            CK(hipMemcpyAsync(got.data(), gout, n * 4, hipMemcpyDeviceToHost, sb));
            CK(hipStreamSynchronize(sb));
            ++launches;
            if (memcmp(got.data(), solo.data(), n * 4) != 0) {
                ++bad;
                for (size_t i = 0; i < n; ++i) if (memcmp(&got[i], &solo[i], 4) != 0) ++lanes[i & 63];
            }
        }
        CK(hipStreamSynchronize(sa));
    }
    printf("%-58s host regs %3d  guest launches %6ld  deviating %5ld  lanes:", label, fa.numRegs, launches, bad);
    for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d:%ld", l, lanes[l]);
    printf("\n");
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const bool role_host = argc > 3 && !strcmp(argv[1], "host"), role_guest = argc > 3 && !strcmp(argv[1], "guest");
    const double seconds = role_host ? atof(argv[3]) : role_guest ? atof(argv[2]) : argc > 1 ? atof(argv[1]) : 6.0;
    const int guest_blocks = 512;
    float *w, *hout, *grid, *gout;
    CK(hipMalloc(&w, 64 * 256 * 16)); CK(hipMalloc(&hout, 256 * 256 * 4)); CK(hipMalloc(&grid, 16384 * 3 * 4));
    const size_t n = (size_t)48 * guest_blocks * 256;
    CK(hipMalloc(&gout, n * 4));
    std::vector<float> init(16384 * 3);
    for (size_t i = 0; i < init.size(); ++i) init[i] = 1e-3f * (float)((i * 2654435761u) >> 22);
    CK(hipMemcpy(grid, init.data(), init.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, init.data(), 64 * 256 * 16, hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    if (role_host) {                                          // this process only runs the host kernel of one mode
        std::vector<float> none(1);
        const int m = atoi(argv[2]);
        return m == 0 ? run_mode<0>("(host process)", seconds, w, hout, grid, gout, none, sa, sb, guest_blocks, true, false)
             : m == 1 ? run_mode<1>("(host process)", seconds, w, hout, grid, gout, none, sa, sb, guest_blocks, true, false)
             : m == 2 ? run_mode<2>("(host process)", seconds, w, hout, grid, gout, none, sa, sb, guest_blocks, true, false)
                      : run_mode<3>("(host process)", seconds, w, hout, grid, gout, none, sa, sb, guest_blocks, true, false);
    }
    hipFuncAttributes ga;
    CK(hipFuncGetAttributes(&ga, reinterpret_cast<const void*>(guest_kernel)));
    if (!role_guest) printf("guest kernel: %d registers, %d blocks x 256 threads; %.0f s per mode; one process, two streams\n", ga.numRegs, guest_blocks, seconds);
    std::vector<float> solo(n), again(n);
    hipLaunchKernelGGL(guest_kernel, dim3(guest_blocks), dim3(256), 0, sb, grid, gout, 64);
    CK(hipMemcpyAsync(solo.data(), gout, n * 4, hipMemcpyDeviceToHost, sb)); CK(hipStreamSynchronize(sb));
    long solo_bad = 0;
    for (int i = 0; i < 200; ++i) {
        hipLaunchKernelGGL(guest_kernel, dim3(guest_blocks), dim3(256), 0, sb, grid, gout, 64);
        CK(hipMemcpyAsync(again.data(), gout, n * 4, hipMemcpyDeviceToHost, sb)); CK(hipStreamSynchronize(sb));
        solo_bad += memcmp(again.data(), solo.data(), n * 4) != 0;
    }
    if (role_guest) return run_mode<0>(argv[3], seconds, w, hout, grid, gout, solo, sa, sb, guest_blocks, false, true);   // (the host is another process)
    printf("%-58s                guest launches %6d  deviating %5ld\n", "guest alone (no host kernel)", 200, solo_bad);
    if (run_mode<0>("host 424 regs, 1 wave/SIMD, epilogue + stream", seconds, w, hout, grid, gout, solo, sa, sb, guest_blocks)) return 1;
    if (run_mode<3>("host 424 regs, 1 wave/SIMD, MFMAs only", seconds, w, hout, grid, gout, solo, sa, sb, guest_blocks)) return 1;
    if (run_mode<2>("host 248 regs, 1 of 2 workgroups/CU resident (narrow tail)", seconds, w, hout, grid, gout, solo, sa, sb, guest_blocks)) return 1;
    if (run_mode<1>("host 512 regs (whole register file claimed)", seconds, w, hout, grid, gout, solo, sa, sb, guest_blocks)) return 1;
    return 0;
}
