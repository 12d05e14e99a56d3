// residency_lds_lab.hip -- LAB (never part of the product build): candidate (b) of DESIGN.md section 8 in its BARE form --
// the activations' cut planes RESIDENT IN LDS (128 samples x 256 features x two fp16 planes = 128 KB, in MFMA B-fragment
// order), eight waves per workgroup (two per SIMD), each on a 64-feature x 64-sample sub-block of the layer's output
// (2 x 2 tiles of v_mfma_f32_32x32x16_f16: 64 accumulator registers), the weights through what LDS is left: a ring of
// 2 x 16 KB = one K slab of all 256 features and both planes per slot, hence ONE WORKGROUP BARRIER PER K SLAB (12 MFMAs per
// wave).  No epilogue, no stores, the activations never change: what is measured is the structure's ceiling -- matrix pipe,
// fragment reads (4 KB of A + 4 KB of B per 12 MFMAs = the 683 B per MFMA of today's residency), the ring and its barrier --
// against the bare builds of the other two residencies (residency_lab.hip with -DSCN_H3_NO_EPI -DSCN_H3_NO_STORE).  A full
// version would add, per layer and workgroup, 128 KB of cut activations written to and 128 KB read back from LDS, a phase
// structure (a sub-block's epilogue can only overwrite its inputs when every wave is done reading them) and a second pass of
// the weight stream for the other half of the samples.
// kind 4 of residency_lab_run (tools/residency_lab.py --kinds lds): the last layer's accumulators go to `save` for the check.
#include <hip/hip_runtime.h>

#include <scn_lab.h>
#include <scn_wave.h>

#include "launch.h"
#include "mlp_h3.h"

namespace lab_lds {

using namespace scn;
using scn::h3::u32x4;

constexpr int kLayers = 8;
constexpr int kThreads = 512;
constexpr unsigned kXBytes = 4 * 16 * 2 * 1024;          // [sample tile 4][slab 16][plane 2] fragments of 1 KB
constexpr unsigned kSlotBytes = 16384;                   // one K slab: [feature tile 8][plane 2] fragments
constexpr unsigned kLds = kXBytes + 2 * kSlotBytes;      // 160 KB: all of it

__host__ __device__ inline float lab_input(unsigned p, unsigned f) {
    unsigned x = p * 1103515245u + f * 12345u + p * f * 7u + 0x9e3779b9u;
    x ^= x >> 15;
    x *= 0x2c1b3c6du;
    x ^= x >> 12;
    return (float)(x >> 8) * (1.f / 16777216.f);
}

__global__ __launch_bounds__(kThreads, 2) void chainlds_kernel(const short* __restrict__ wstream, float* __restrict__ save, long P) {
    char* const lds = dynamic_lds<char>();
    const int lane = lane_id(), m = lane & 31, g = lane >> 5;
    const int wv = uniform(wave_id());
    const int fb = wv >> 1, sb = wv & 1;                 // feature block (64 features), sample block (64 samples)
    const unsigned lane16 = (unsigned)lane * 16u, tid16 = threadIdx.x * 16u;
    const float s0 = h3::scale_for(1.f);
    // the activations: fragment (j, s, plane) holds, for lane (m, g), elements e = 0 .. 7: feature 32 (s >> 1) + 16 (s & 1) +
    // 8 (e >> 2) + 4 g + (e & 3) of sample 32 j + m (mlp_layout.h3_feature_of: today's k order)
    for (int fr = wv; fr < 4 * 16; fr += 8) {
        const int j = fr >> 4, s = fr & 15;
        const unsigned p = (unsigned)(blockIdx.x * 128 + 32 * j + m);
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = lab_input(p, (unsigned)(32 * (s >> 1) + 16 * (s & 1) + 8 * (e >> 2) + 4 * g + (e & 3)));
        u32x4 xh, xl;
        h3::cut8(x, s0, xh, xl);
        *reinterpret_cast<u32x4*>(lds + ((j * 16 + s) * 2 + 0) * 1024 + lane16) = xh;
        *reinterpret_cast<u32x4*>(lds + ((j * 16 + s) * 2 + 1) * 1024 + lane16) = xl;
    }
    // the ring: slab 0 -> slot 0 now, slab 1 -> staging registers
    global_bytes gsrc = uniform_global(wstream);
    f32x4 stage[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(lds + kXBytes + i * 8192 + tid16) = load_f32x4(gsrc + i * 8192, tid16);
    gsrc = uniform_global(gsrc + kSlotBytes);
#pragma unroll
    for (int i = 0; i < 2; ++i) stage[i] = load_f32x4(gsrc + i * 8192, tid16);
    gsrc = uniform_global(gsrc + kSlotBytes);
    block_sync();

    f32x16 acc[2][2];
    unsigned slot = 0u;
#pragma unroll 1
    for (int layer = 0; layer < kLayers; ++layer) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            // this slab's fragments: A (this wave's two feature tiles, both planes) from the ring, B (its two sample tiles) from X
            s16x8 a[2][2], b[2][2];
            const char* ring = lds + kXBytes + slot;
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    a[ft][pl] = *reinterpret_cast<const s16x8*>(ring + ((2 * fb + ft) * 2 + pl) * 1024 + lane16);
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    b[st][pl] = *reinterpret_cast<const s16x8*>(lds + (((2 * sb + st) * 16 + s) * 2 + pl) * 1024 + lane16);
            // the next slab goes from its staging registers into the other slot (every wave left that slot at the last barrier),
            // and the registers are refilled with the slab after it
            if constexpr (!lab::kNoStream) {
#pragma unroll
                for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(lds + kXBytes + (slot ^ kSlotBytes) + i * 8192 + tid16) = stage[i];
#pragma unroll
                for (int i = 0; i < 2; ++i) stage[i] = load_f32x4(uniform_global(gsrc + i * 8192), pinned_here(tid16));
                gsrc = uniform_global(gsrc + kSlotBytes);
            }
            sched_fence();
            // (Wh Xh), (Wh Xl), (Wl Xh): product-major, so consecutive MFMAs never target the same accumulator
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const s16x8 aa = a[ft][pr == 2 ? 1 : 0], bb = b[st][pr == 1 ? 1 : 0];
                        if (s == 0 && pr == 0) {
                            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[ft][st] = mfma_32x32x16_f16(aa, bb, zero);
                        } else {
                            acc[ft][st] = mfma_32x32x16_f16(aa, bb, acc[ft][st]);
                        }
                    }
            sched_fence();
            if constexpr (!lab::kNoBarrier) block_sync();
            slot ^= kSlotBytes;
        }
    }
    // the last layer's accumulators: [workgroup][wave][ft][st][register 16][lane 64]
    float* out = save + ((long)blockIdx.x * 8 + wv) * (4 * 16 * 64);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[((ft * 2 + st) * 16 + r) * 64 + lane] = acc[ft][st][r];
    (void)P;
}

}  // namespace lab_lds

extern "C" int residency_lds_lab_run(const void* wstream, void* save, long long P, int reps, float* ms) {
    using namespace lab_lds;
    SCN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chainlds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
    const dim3 grid(scn_ceil_div(P, 128));
#ifndef SCNERF_SIMT_EMU_BUILD
    hipEvent_t e0, e1;
    SCN_HIP(hipEventCreate(&e0));
    SCN_HIP(hipEventCreate(&e1));
    for (int r = 0; r < reps; ++r) {
        SCN_HIP(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(chainlds_kernel, grid, dim3(kThreads), kLds, nullptr, static_cast<const short*>(wstream), static_cast<float*>(save), (long)P);
        SCN_HIP(hipEventRecord(e1, nullptr));
        SCN_HIP(hipEventSynchronize(e1));
        SCN_HIP(hipEventElapsedTime(&ms[r], e0, e1));
    }
    SCN_HIP(hipEventDestroy(e0));
    SCN_HIP(hipEventDestroy(e1));
#else
    (void)reps;
    hipLaunchKernelGGL(chainlds_kernel, grid, dim3(kThreads), kLds, nullptr, static_cast<const short*>(wstream), static_cast<float*>(save), (long)P);
    ms[0] = 0.f;
#endif
    return scn_launch_status();
}
