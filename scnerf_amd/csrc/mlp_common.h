// mlp_common.h -- pieces shared by the fused MLP forward / dgrad kernels.
//
// Register-resident activations (see scnerf_amd/mlp_layout.py and DESIGN.md): a wave owns 32
// samples; lane (m = l & 31, h = l >> 5) holds, for sample m, feature
//     feat_of(t, r, h) = 32 t + (r & 3) + 8 (r >> 2) + 4 h
// in element r of accumulator tile t (the D layout of v_mfma_f32_32x32x2_f32 when the
// product is computed transposed, D[feature][sample] = W[feature][k] * X[k][sample]).
// MFMA step s = 16 t + r of the next layer contracts the feature pair held in register
// (t, r) by the two lane halves, so the previous layer's output registers ARE the next
// layer's B operands.  The A operand (weights, pre-gathered into consumption order by
// scnerf_gather_f32) streams L2 -> LDS in double-buffered chunks shared by the 4 waves of
// a workgroup and is read with ds_read_b128 (4 consecutive steps per lane).
#pragma once
#include <scn_wave.h>

namespace scn {
namespace mlp {

constexpr int kThreads = 256;           // 4 waves, one per SIMD (the kernel needs ~400 VGPRs)
constexpr int kSamplesPerWave = 32;
constexpr int kSamplesPerBlock = 128;

// ---- packed parameter buffer offsets (floats); mirrored in scnerf_amd/mlp_layout.py ----
constexpr int kFwdStream = 598016;
constexpr int kFwdBias = kFwdStream;          // 8 trunk layers x 256, half-pair layout
constexpr int kFwdBiasF = kFwdBias + 8 * 256;
constexpr int kFwdBiasV = kFwdBiasF + 256;
constexpr int kFwdBiasRGB = kFwdBiasV + 128;
constexpr int kFwdAlphaW = kFwdBiasRGB + 32;
constexpr int kFwdAlphaB = kFwdAlphaW + 256;
constexpr int kFwdTotal = kFwdAlphaB + 4;
constexpr int kBwdStream = 1024 + 64 * 9 * 64 + 3 * 65536 + 128 * 10 * 64 + 4 * 65536 + 128 * 2 * 64;
constexpr int kBwdAlphaW = kBwdStream;
constexpr int kBwdTotal = kBwdAlphaW + 256;

constexpr int kMaxChunkFwd = 8192;      // floats: 32 KB, x2 buffers = 64 KB LDS
constexpr int kMaxChunkBwd = 8192;      // floats: 32 KB, x2 buffers = 64 KB LDS

// activation / gradient workspace section widths (row-major [P][width])
constexpr int kSaveAct = 0;             // 8 x [P][256]
constexpr int kSaveFeat = 8 * 256;      // offsets in floats-per-sample units (multiply by P)
constexpr int kSaveHv = kSaveFeat + 256;
constexpr int kSaveEpts = kSaveHv + 128;
constexpr int kSaveEviews = kSaveEpts + 64;
constexpr int kSavePerSample = kSaveEviews + 32;   // 2592
// after the row sections: ReLU bit masks, lane-native: [9 sections][wave tile][64 lanes][4 words]
// (sections 0..7 = trunk layers, 8 = views layer); 8 words per (padded) sample and section.
constexpr int kMaskSections = 9;
constexpr int kMaskWordsPerSample = kMaskSections * 8;   // 72
constexpr int kGradDz = 0;              // 8 x [P][256]
constexpr int kGradDfeat = 8 * 256;
constexpr int kGradDzv = kGradDfeat + 256;
constexpr int kGradPerSample = kGradDzv + 128;     // 2432

__host__ __device__ constexpr int feat_of(int t, int r, int h) {
    return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// Embedding column of PE slot s on lane-half h (torch order: x, then per frequency
// [sin xyz, cos xyz]); -1 = zero pad.  Mirrors mlp_layout.pe_col.
__host__ __device__ constexpr int pe_col(int L, int s, int h) {
    if (s < 3 * L) {
        const int f = s / 3, j = s % 3;
        if (j == 0) return 3 + 6 * f + h;
        if (j == 1) return 3 + 6 * f + 3 + h;
        return 3 + 6 * f + (h == 0 ? 2 : 5);
    }
    if (s == 3 * L) return h;
    if (s == 3 * L + 1) return h == 0 ? 2 : -1;
    return -1;
}

// Positional encoding of (x, y, z) straight into MFMA B-operand registers: NS slots.
// Slots 3f, 3f+1 = sin / cos of (h ? y : x) * 2^f; slot 3f+2 = (h ? cos : sin)(z * 2^f);
// slot 3L = raw x|y; slot 3L+1 = raw z | 0.  x * 2^f is exact, as in the reference.
template <int L, int NS>
__device__ __forceinline__ void pe_slots(float x, float y, float z, int h, float (&e)[NS]) {
    const float xy = h ? y : x;
    float freq = 1.f;
#pragma unroll
    for (int f = 0; f < L; ++f) {
        float s0, c0, s1, c1;
        sincos(xy * freq, &s0, &c0);
        sincos(z * freq, &s1, &c1);
        e[3 * f + 0] = s0;
        e[3 * f + 1] = c0;
        e[3 * f + 2] = h ? c1 : s1;
        freq *= 2.f;
    }
    e[3 * L] = xy;
    e[3 * L + 1] = h ? 0.f : z;
#pragma unroll
    for (int s = 3 * L + 2; s < NS; ++s) e[s] = 0.f;
}

// ---- weight stream: global (L2) -> LDS, one chunk ahead of the MFMAs ---------------------
struct WStream {
    const f32x4* g;      // next chunk to fetch
    float* buf[2];       // LDS double buffer
    int cur;             // buffer holding the chunk being consumed
};

// All counts are compile-time so the staging registers stay registers.
template <int N_F4>
__device__ __forceinline__ void stream_issue(const WStream& ws, f32x4 (&stage)[N_F4 > 0 ? N_F4 : 1]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < N_F4; ++i) stage[i] = ws.g[i * kThreads + tid];
}

template <int N_F4>
__device__ __forceinline__ void stream_commit(WStream& ws, const f32x4 (&stage)[N_F4 > 0 ? N_F4 : 1]) {
    const int tid = threadIdx.x;
    f32x4* dst = reinterpret_cast<f32x4*>(ws.buf[ws.cur ^ 1]);
#pragma unroll
    for (int i = 0; i < N_F4; ++i) dst[i * kThreads + tid] = stage[i];
    ws.g += N_F4 * kThreads;
}

// First chunk of the whole stream (kernel prologue).
template <int N_F4>
__device__ __forceinline__ void stream_prime(WStream& ws) {
    f32x4 stage[N_F4];
    ws.cur = 1;                        // commit() writes buf[cur ^ 1] = buf[0]
    stream_issue<N_F4>(ws, stage);
    stream_commit<N_F4>(ws, stage);
    ws.cur = 0;
    block_sync();
}

// One chunk: CS steps x NT tiles out of LDS buffer `A`, B operands b[B0 .. B0 + CS).
template <int NSTEP, int NT, int CS, int B0>
__device__ __forceinline__ void mfma_chunk(const float (&b)[NSTEP], f32x16 (&acc)[NT], const f32x4* A,
                                           int lane) {
    constexpr int G = CS / 4;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 a = A[(t * G + g) * 64 + lane];
            acc[t] = mfma_32x32x2(a[0], b[B0 + 4 * g + 0], acc[t]);
            acc[t] = mfma_32x32x2(a[1], b[B0 + 4 * g + 1], acc[t]);
            acc[t] = mfma_32x32x2(a[2], b[B0 + 4 * g + 2], acc[t]);
            acc[t] = mfma_32x32x2(a[3], b[B0 + 4 * g + 3], acc[t]);
        }
    }
}

template <int NSTEP, int NT, int CS, int NEXT_F4, int C>
struct PartLoop {
    static constexpr int NC = NSTEP / CS;
    static constexpr int CHUNK_F4 = NT * CS * 64 / 4 / kThreads;
    static constexpr int N_F4 = (C + 1 < NC) ? CHUNK_F4 : NEXT_F4;
    static __device__ __forceinline__ void run(const float (&b)[NSTEP], f32x16 (&acc)[NT], WStream& ws,
                                               int lane) {
        f32x4 stage[N_F4 > 0 ? N_F4 : 1];
        stream_issue<N_F4>(ws, stage);
        mfma_chunk<NSTEP, NT, CS, C * CS>(b, acc, reinterpret_cast<const f32x4*>(ws.buf[ws.cur]), lane);
        stream_commit<N_F4>(ws, stage);
        block_sync();
        ws.cur ^= 1;
        if constexpr (C + 1 < NC) PartLoop<NSTEP, NT, CS, NEXT_F4, C + 1>::run(b, acc, ws, lane);
    }
};

// One "part" = NSTEP MFMA steps over NT output tiles with B operands taken from registers
// b[0..NSTEP).  CS steps per LDS chunk; NEXT_F4 = 16-byte loads per thread of the chunk that
// follows this part in the stream (0 at the end of the stream).
template <int NSTEP, int NT, int CS, int NEXT_F4>
__device__ __forceinline__ void mfma_part(const float (&b)[NSTEP], f32x16 (&acc)[NT], WStream& ws) {
    static_assert(NSTEP % CS == 0 && CS % 4 == 0, "chunking");
    static_assert((NT * CS * 64) % (4 * kThreads) == 0, "chunk must be whole 16-byte loads per thread");
    PartLoop<NSTEP, NT, CS, NEXT_F4, 0>::run(b, acc, ws, lane_id());
}

// acc[t][r] = bias of feature feat_of(t, r, h); bias_hp is the half-pair table
// [(16 t + r) * 2 + h] (wave-uniform address -> scalar loads + one select per register).
template <int NT>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NT], const float* __restrict__ bias_hp, int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float b0 = bias_hp[(16 * t + r) * 2 + 0];
            const float b1 = bias_hp[(16 * t + r) * 2 + 1];
            acc[t][r] = h ? b1 : b0;
        }
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// rows [p][col0 + feat_of(t, 4q .. 4q+3, h)] <- 4 consecutive features per 16-byte store
template <int NT>
__device__ __forceinline__ void store_rows(const float* regs, float* __restrict__ base, long p,
                                           int ld, int h, bool live) {
    if (!live) return;
    float* row = base + p * ld + 4 * h;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {regs[16 * t + 4 * q + 0], regs[16 * t + 4 * q + 1], regs[16 * t + 4 * q + 2],
                       regs[16 * t + 4 * q + 3]};
            *reinterpret_cast<f32x4*>(row + 32 * t + 8 * q) = v;
        }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline long padded_samples(long P) {
    return (P + kSamplesPerBlock - 1) / kSamplesPerBlock * kSamplesPerBlock;
}

// bit i of the lane's mask = (v[i] > 0); N <= 128 registers -> 4 words
template <int N>
__device__ __forceinline__ u32x4 relu_bits(const float (&v)[N]) {
    u32x4 bits = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < N; ++i) bits[i >> 5] |= (v[i] > 0.f ? 1u : 0u) << (i & 31);
    return bits;
}

__device__ __forceinline__ unsigned int* mask_ptr(float* save, long P, int section, long wave_tile, int lane) {
    unsigned int* base = reinterpret_cast<unsigned int*>(save + (long)kSavePerSample * P);
    return base + ((long)section * (padded_samples(P) / 32) + wave_tile) * 256 + lane * 4;
}

}  // namespace mlp
}  // namespace scn
