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
#include <type_traits>

#include <scn_lab.h>
#include <scn_wave.h>

namespace scn {
namespace mlp {

constexpr int kThreads = 256;           // 4 waves, one per SIMD (the kernel needs ~400 VGPRs)
constexpr int kSamplesPerWave = 32;
constexpr int kSamplesPerBlock = 128;

// ---- network variants ---------------------------------------------------------------------------
// PD = dimensions of the encoded point: 3 = (x, y, z), the SCNeRF / NeRF++-foreground network
// (63 encoded columns, 32 PE slots per lane half); 4 = (x, y, z, 1/r), the NeRF++ background network
// (nerfplusplus/ddp_model.py:62-71: 84 columns, 48 slots).  Everything else -- trunk, skip, heads -- is
// identical, so the two variants share every part of the weight stream except the encoded-point ones.
template <int PD>
struct Var {
    static_assert(PD == 3 || PD == 4, "point dimensions");
    static constexpr int kInCh = PD + 2 * PD * 10;           // 63 / 84 encoded columns
    static constexpr int kES = PD == 3 ? 32 : 48;            // PE slots (MFMA steps) of the encoded point
    static constexpr int kEW = PD == 3 ? 64 : 128;           // row width of the saved encodings (wgrad K)
    static constexpr int kET = PD == 3 ? 2 : 4;              // 32-column tiles of the dgrad's d-encoding
    static constexpr int kECS = PD == 3 ? 64 : 32;           // dgrad chunk steps of those parts (8192-float chunks)
    // ---- packed parameter buffer offsets (floats); mirrored in scnerf_amd/mlp_layout.py ----
    static constexpr int kFwdStream = 2 * kES * 8 * 64 + 8 * 65536 + 128 * 4 * 64 + 16 * 4 * 64 + 64 * 64;
    static constexpr int kFwdBias = kFwdStream;              // 8 trunk layers x 256, half-pair layout
    static constexpr int kFwdBiasF = kFwdBias + 8 * 256;
    static constexpr int kFwdBiasV = kFwdBiasF + 256;
    static constexpr int kFwdBiasRGB = kFwdBiasV + 128;
    static constexpr int kFwdAlphaW = kFwdBiasRGB + 32;
    static constexpr int kFwdAlphaB = kFwdAlphaW + 256;
    static constexpr int kFwdTotal = kFwdAlphaB + 4;
    static constexpr int kBwdStream = 1024 + 64 * 9 * 64 + 8 * 65536 + 2 * 128 * kET * 64;
    static constexpr int kBwdAlphaW = kBwdStream;
    static constexpr int kBwdTotal = kBwdAlphaW + 256;
    static constexpr int kSavePerSample = 8 * 256 + 256 + 128 + 32 + kEW;
    // flat parameter offsets, reference registration order (mirrors mlp_layout.Layout.param_offsets)
    static constexpr int kW0 = 0, kB0 = kW0 + 256 * kInCh;
    static constexpr int kTrunk1 = kB0 + 256;                // layers 1..7: weight then bias
    static constexpr int kSkipLd = 256 + kInCh;
    static constexpr int trunk_w(int l) {
        return l <= 5 ? kTrunk1 + (l - 1) * (256 * 256 + 256)
                      : kTrunk1 + 4 * (256 * 256 + 256) + (256 * kSkipLd + 256) + (l - 6) * (256 * 256 + 256);
    }
    static constexpr int trunk_b(int l) { return trunk_w(l) + (l == 5 ? 256 * kSkipLd : 256 * 256); }
    static constexpr int kWV = trunk_b(7) + 256, kBV = kWV + 128 * 283;
    static constexpr int kWF = kBV + 128, kBF = kWF + 256 * 256;
    static constexpr int kWA = kBF + 256, kBA = kWA + 256;
    static constexpr int kWRGB = kBA + 1, kBRGB = kWRGB + 3 * 128;
    static constexpr int kNParams = kBRGB + 3;
};
static_assert(Var<3>::kFwdStream == 598016 && Var<3>::kBwdStream == 594944 && Var<3>::kNParams == 595844, "standard network");
static_assert(Var<4>::kNParams == 606596, "NeRF++ background network");

constexpr int kMaxChunkFwd = 8192;      // floats: 32 KB, x3 buffers = 96 KB LDS
constexpr int kMaxChunkBwd = 8192;      // floats: 32 KB, x3 buffers = 96 KB LDS
constexpr int kStreamBufs = 3;
constexpr int kInterleave = 2;          // accumulator tiles interleaved in the fp32 kernels' MFMA order (4-way measured: forward -0.3 %, dgrad +1.8 %; 2 kept)
static_assert(kMaxChunkFwd == 8192 && kMaxChunkBwd == 8192, "WStream::buf stride");

// Activation / gradient workspaces.  Offsets are in floats per (padded) sample: a section starts at
// offset * padded_samples(P).  The wide sections (act*, feat, hv, dz*, dfeat, dzv) are TILE-NATIVE:
// per wave tile of 32 samples a block [t][q][lane][4] (t = 32-feature tile, q = 0..3, lane = (m, h)),
// holding feature 32 t + 8 q + 4 h + j of sample m at element j -- exactly the registers a lane owns,
// so every store instruction of the MLP kernels writes 1 KB contiguous.  The wgrad GEMM stages whole
// tiles linearly into LDS and reads this layout there.  eviews / epts are row-major [P][32] / [P][kEW]
// (epts last: its width is the only variant-dependent one).
constexpr int kSaveAct = 0;             // 8 sections of width 256
constexpr int kSaveFeat = 8 * 256;
constexpr int kSaveHv = kSaveFeat + 256;
constexpr int kSaveEviews = kSaveHv + 128;
constexpr int kSaveEpts = kSaveEviews + 32;
// after the row sections (Var<PD>::kSavePerSample floats per sample): ReLU bit masks, lane-native:
// [9 sections][wave tile][64 lanes][4 words] (sections 0..7 = trunk layers, 8 = views layer); 8 words
// per (padded) sample and section.
constexpr int kMaskSections = 9;
constexpr int kMaskWordsPerSample = kMaskSections * 8;   // 72
constexpr int kGradDz = 0;              // 8 x [P][256]
constexpr int kGradDfeat = 8 * 256;
constexpr int kGradDzv = kGradDfeat + 256;
constexpr int kGradPerSample = kGradDzv + 128;     // 2432

__host__ __device__ constexpr int feat_of(int t, int r, int h) {
    return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// Embedding column of PE slot s on lane-half h (torch order: the raw coordinates, then per frequency
// [sin of each, cos of each]); -1 = zero pad.  Mirrors mlp_layout.pe_col.
//   PD == 3: slots 3f, 3f+1 = sin / cos of (h ? y : x) 2^f; slot 3f+2 = (h ? cos : sin)(z 2^f);
//            slot 3L = raw x|y; slot 3L+1 = raw z | pad.
//   PD == 4: half h owns coordinates (h ? y : x) and (h ? w : z): slots 4f .. 4f+3 = sin, cos of the
//            first, sin, cos of the second; slots 4L, 4L+1 = the raw pair.
__host__ __device__ constexpr int pe_col(int L, int s, int h, int PD = 3) {
    if (PD == 3) {
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
    if (s < 4 * L) {
        const int f = s / 4, j = s % 4;
        const int coord = (j < 2 ? 0 : 2) + h;              // x|y then z|w
        return 4 + 8 * f + ((j & 1) ? 4 : 0) + coord;       // sin block then cos block of frequency f
    }
    if (s == 4 * L) return h;
    if (s == 4 * L + 1) return 2 + h;
    return -1;
}

// Positional encoding of a point straight into MFMA B-operand registers: NS slots (layout above).
// x * 2^f is exact, as in the reference.  `w` is ignored for PD == 3.
template <int PD, int L, int NS>
__device__ __forceinline__ void pe_slots(float x, float y, float z, float w, int h, float (&e)[NS]) {
    const float a = h ? y : x;
    float freq = 1.f;
    if constexpr (PD == 3) {
#pragma unroll
        for (int f = 0; f < L; ++f) {
            float s0, c0, s1, c1;
            sincos(a * freq, &s0, &c0);
            sincos(z * freq, &s1, &c1);
            e[3 * f + 0] = s0;
            e[3 * f + 1] = c0;
            e[3 * f + 2] = h ? c1 : s1;
            freq *= 2.f;
        }
        e[3 * L] = a;
        e[3 * L + 1] = h ? 0.f : z;
#pragma unroll
        for (int s = 3 * L + 2; s < NS; ++s) e[s] = 0.f;
    } else {
        const float b = h ? w : z;
#pragma unroll
        for (int f = 0; f < L; ++f) {
            sincos(a * freq, &e[4 * f + 0], &e[4 * f + 1]);
            sincos(b * freq, &e[4 * f + 2], &e[4 * f + 3]);
            freq *= 2.f;
        }
        e[4 * L] = a;
        e[4 * L + 1] = b;
#pragma unroll
        for (int s = 4 * L + 2; s < NS; ++s) e[s] = 0.f;
    }
}

// ---- weight stream: global (L2) -> LDS, one chunk ahead of the MFMAs ---------------------
// THREE LDS buffers.  While chunk c is consumed from buf[c % 3], the staged registers of chunk c+1 are
// written to buf[(c+1) % 3] half-way through the chunk and the workgroup barrier sits at three quarters:
// behind it chunk c+1 is complete, so its first A fragments are fetched during the tail of chunk c and no
// chunk boundary ever waits for LDS (with two buffers the barrier had to sit AT the boundary: ~250 idle
// cycles per 8192-cycle chunk).  The third buffer is what makes the early write safe: buf[(c+1) % 3] was
// last read as chunk c-2, which every wave had finished when it passed the barrier inside chunk c-1.
struct WStream {
    const f32x4* g;      // next chunk to fetch
    int cur;             // LDS buffer (0..2) holding the chunk being consumed
    __device__ __forceinline__ int next() const { return cur == 2 ? 0 : cur + 1; }
    // Buffers are addressed as offsets from the dynamic-LDS symbol, never through stored pointers: a
    // pointer selected at run time loses its address space and the reads degrade to FLAT loads.
    static __device__ __forceinline__ float* buf(int i) { return dynamic_lds<float>() + i * 8192; }
};

// a 16-byte store to the activation / gradient workspace: written once, read by a LATER kernel, 7-8 GB per
// launch -- non-temporal, so the stream does not displace the packed weights every workgroup re-reads from L2
// (dgrad -0.8 %, forward unchanged)
__device__ __forceinline__ void store_ws(f32x4* p, f32x4 v) {
    if constexpr (lab::kPlainStore) *p = v;           // (timing experiment: default cache policy)
    else __builtin_nontemporal_store(v, p);
}

// All counts are compile-time so the staging registers stay registers.
template <int N_F4>
__device__ __forceinline__ void stream_issue(const WStream& ws, f32x4 (&stage)[N_F4 > 0 ? N_F4 : 1]) {
    const int tid = threadIdx.x;
    // (scalar-base addressing of these loads -- readfirstlane'd base + 32-bit lane offset -- removes the
    // 64-bit VALU address adds but measured 1.5 % slower; plain pointer arithmetic kept)
#pragma unroll
    for (int i = 0; i < N_F4; ++i) stage[i] = ws.g[i * kThreads + tid];
}

template <int N_F4>
__device__ __forceinline__ void stream_commit(WStream& ws, const f32x4 (&stage)[N_F4 > 0 ? N_F4 : 1]) {
    const int tid = threadIdx.x;
    f32x4* dst = reinterpret_cast<f32x4*>(WStream::buf(ws.next()));
#pragma unroll
    for (int i = 0; i < N_F4; ++i) dst[i * kThreads + tid] = stage[i];
    ws.g += N_F4 * kThreads;
}

// ---- spread schedule -----------------------------------------------------------------------------------
// A chunk is walked in NB bundles of 8 MFMAs.  Issued as bursts (all loads at the head, all LDS writes
// half-way, all activation stores at the head) the memory instructions of the four waves collide: each wave
// sits in front of a full queue while its matrix pipe drains (measured on the weight-gradient GEMM, which has
// the same structure: 119 -> 140 TFLOP/s by spreading, profiles/r02_wgrad_lab_*.txt).  Spread:
//   bundles [0, NB/4)       the global loads of the NEXT chunk,
//   bundles [NB/4, 3NB/4)   its LDS writes, one or two per bundle, in load order,
//   bundle  3NB/4           the workgroup barrier (unchanged),
//   bundles [3NB/4, NB)     the activation / gradient stores of this chunk's B operands (training).
// Chunks with fewer than 8 bundles keep the burst form.
template <int NB, int N>
struct Spread {
    static constexpr bool kOn = !lab::kBurst && NB >= 8 && N > 0;       // (lab::kBurst: the round-1 burst schedule)
    static constexpr int kQ = NB / 4;
    static constexpr int load_at(int i) { return (i * kQ) / (N > 0 ? N : 1); }
    static constexpr int commit_at(int i) { return kQ + (i * 2 * kQ) / (N > 0 ? N : 1); }
    static constexpr int store_at(int k, int n_store) { return 3 * kQ + (k * (NB - 3 * kQ)) / n_store; }
};

template <int N_F4>
__device__ __forceinline__ void stream_issue_piece(const WStream& ws, f32x4 (&stage)[N_F4 > 0 ? N_F4 : 1], int i) {
    stage[i] = ws.g[i * kThreads + threadIdx.x];
}

template <int N_F4>
__device__ __forceinline__ void stream_commit_piece(const WStream& ws, const f32x4 (&stage)[N_F4 > 0 ? N_F4 : 1], int i) {
    f32x4* dst = reinterpret_cast<f32x4*>(WStream::buf(ws.next()));
    dst[i * kThreads + threadIdx.x] = stage[i];
}

// store k (= (t - T0) * 4 + q) of store_tiles<T0, T1>
template <int T0, int N>
__device__ __forceinline__ void store_tile_piece(const float (&regs)[N], float* tile, int k) {
    if (tile == nullptr) return;
    const int t = T0 + k / 4, q = k % 4;
    f32x4 v = {regs[16 * t + 4 * q + 0], regs[16 * t + 4 * q + 1], regs[16 * t + 4 * q + 2], regs[16 * t + 4 * q + 3]};
    store_ws(reinterpret_cast<f32x4*>(tile + (t * 4 + q) * 256), v);
}

// everything the spread schedule does in bundle p of a chunk (compile-time p after unrolling)
template <int NB, int N_F4, int NSTEP, int T0, int T1>
__device__ __forceinline__ void spread_bundle(int p, WStream& ws, f32x4 (&stage)[N_F4 > 0 ? N_F4 : 1],
                                              const float (&b)[NSTEP], float* save_tile) {
    using S = Spread<NB, N_F4>;
    if constexpr (S::kOn) {
#pragma unroll
        for (int i = 0; i < N_F4; ++i)
            if (S::load_at(i) == p) stream_issue_piece<N_F4>(ws, stage, i);
#pragma unroll
        for (int i = 0; i < N_F4; ++i)
            if (S::commit_at(i) == p) stream_commit_piece<N_F4>(ws, stage, i);
    }
    if constexpr (Spread<NB, 1>::kOn && T1 > T0) {
        constexpr int NS = (T1 - T0) * 4;
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (Spread<NB, 1>::store_at(k, NS) == p) store_tile_piece<T0, NSTEP>(b, save_tile, k);
    }
}

// First chunk of the whole stream (kernel prologue).
template <int N_F4>
__device__ __forceinline__ void stream_prime(WStream& ws) {
    f32x4 stage[N_F4];
    ws.cur = 2;                        // commit() writes buf[next()] = buf[0]
    stream_issue<N_F4>(ws, stage);
    stream_commit<N_F4>(ws, stage);
    ws.cur = 0;
    block_sync();
}

// tile-native block <- registers 16 t + 4 q + (0..3): one 16-byte store per (t, q), lane-contiguous.
// `tile` points at this lane's 16 bytes inside (t, q) = (0, 0) of its wave tile's block (or is nullptr
// when nothing is saved).  Lanes without a sample store too (their slot exists; values are finite).
template <int T0, int T1, int N>
__device__ __forceinline__ void store_tiles(const float (&regs)[N], float* tile) {
    if (tile == nullptr) return;
#pragma unroll
    for (int t = T0; t < T1; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {regs[16 * t + 4 * q + 0], regs[16 * t + 4 * q + 1], regs[16 * t + 4 * q + 2],
                       regs[16 * t + 4 * q + 3]};
            store_ws(reinterpret_cast<f32x4*>(tile + (t * 4 + q) * 256), v);     // 64 lanes x 16 B contiguous
        }
}

// One chunk: CS steps x NT tiles out of LDS buffer buf[cur], B operands b[B0 .. B0 + CS).
// The A fragments (one ds_read_b128 = 4 consecutive steps of one tile) go through a register ring with
// prefetch distance one pair -- 512 MFMA cycles ahead of use, so LDS latency never reaches the matrix pipe
// (the compiler's own schedule kept one read in flight and waited on it: ~20 % idle).  Half-way the staged
// registers of the NEXT chunk are written to the next LDS buffer, at three quarters the workgroup
// synchronises; CONT_IN: the ring already holds this chunk's first pair (fetched by the previous chunk);
// CONT_OUT: fetch the next chunk's first pair from the next buffer (same part: same fragment geometry).
template <int NSTEP, int NT, int CS, int B0, int N_F4, bool CONT_IN, bool CONT_OUT, int T0 = 0, int T1 = 0>
__device__ __forceinline__ void mfma_chunk(const float (&b)[NSTEP], f32x16 (&acc)[NT], WStream& ws,
                                           f32x4 (&stage)[N_F4 > 0 ? N_F4 : 1], f32x4 (&ring)[8], int lane,
                                           float* save_tile = nullptr) {
    constexpr int G = CS / 4;
    constexpr int NF = G * NT;                 // fragments, order f = g * NT + t
    const f32x4* A = reinterpret_cast<const f32x4*>(WStream::buf(ws.cur)) + lane;
    if constexpr (NT % 2 == 0) {
        // IW tiles are interleaved so that consecutive MFMAs never target the same accumulator: a dependent
        // chain on one accumulator issues slower than 64 cycles (tools/ubench/mfma_dep.hip: 145 / 148 / 151 /
        // 154 TFLOP/s with 1 / 2 / 4 / 8 independent chains).
        constexpr int IW = (NT % 4 == 0) ? kInterleave : 2;
        constexpr int NB = NF / IW;            // fragment bundles (tiles t .. t+IW-1 of the same 4 steps)
        static_assert(NB % 2 == 0 || !(CONT_IN || CONT_OUT), "ring parity across chunks");
        // the next chunk's LDS writes half-way through the chunk, the barrier at three quarters
        constexpr int COMMIT_AT = (NB * 2) / 4;
        constexpr int SYNC_AT = (NB * 3) / 4 >= NB ? NB - 1 : (NB * 3) / 4;
        const f32x4* An = reinterpret_cast<const f32x4*>(WStream::buf(ws.next())) + lane;
        constexpr bool SPREAD = Spread<NB, N_F4>::kOn;
        static_assert(!SPREAD || (SYNC_AT == 3 * (NB / 4) && Spread<NB, N_F4>::commit_at(N_F4 - 1) < SYNC_AT),
                      "every LDS write of the next chunk is issued before the barrier");
        if constexpr (!SPREAD) {
            stream_issue<N_F4>(ws, stage);
            sched_fence();      // the loads stay at the head of the chunk: half a chunk of MFMAs covers them
        }
        if constexpr (!Spread<NB, 1>::kOn) store_tiles<T0, T1, NSTEP>(b, save_tile);
        if constexpr (!CONT_IN) {
#pragma unroll
            for (int k = 0; k < IW; ++k) ring[k] = A[((k % NT) * G + k / NT) * 64];
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            spread_bundle<NB, N_F4, NSTEP, T0, T1>(p, ws, stage, b, save_tile);
            if (!SPREAD && p == COMMIT_AT) stream_commit<N_F4>(ws, stage);
            if (p == SYNC_AT) {
                block_sync();
            }
            if (p + 1 < NB) {
#pragma unroll
                for (int k = 0; k < IW; ++k) {
                    const int f = IW * (p + 1) + k;
                    ring[IW * ((p + 1) & 1) + k] = A[((f % NT) * G + f / NT) * 64];
                }
            } else if constexpr (CONT_OUT) {
#pragma unroll
                for (int k = 0; k < IW; ++k) ring[k] = An[((k % NT) * G + k / NT) * 64];
            }
            sched_fence();      // keep the reads one bundle ahead of their use
            const int f = IW * p, g = f / NT, t = f % NT;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < IW; ++k)
                    acc[t + k] = mfma_32x32x2(ring[IW * (p & 1) + k][j], b[B0 + 4 * g + j], acc[t + k]);
        }
        if constexpr (SPREAD) ws.g += N_F4 * kThreads;
    } else {
        static_assert(!CONT_IN && !CONT_OUT, "single-tile parts do not chain");
        stream_issue<N_F4>(ws, stage);
        sched_fence();
        store_tiles<T0, T1, NSTEP>(b, save_tile);
        constexpr int COMMIT_AT = NF / 2;
        constexpr int SYNC_AT = (NF * 3) / 4;
        f32x4 r3[3];
        r3[0] = A[((0 % NT) * G + 0 / NT) * 64];
        if (NF > 1) r3[1] = A[((1 % NT) * G + 1 / NT) * 64];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (f == COMMIT_AT) stream_commit<N_F4>(ws, stage);
            if (f == SYNC_AT) {
                block_sync();
            }
            if (f + 2 < NF) r3[(f + 2) % 3] = A[(((f + 2) % NT) * G + (f + 2) / NT) * 64];
            sched_fence();
            const int g = f / NT, t = f % NT;
            const f32x4 a = r3[f % 3];
            acc[t] = mfma_32x32x2(a[0], b[B0 + 4 * g + 0], acc[t]);
            acc[t] = mfma_32x32x2(a[1], b[B0 + 4 * g + 1], acc[t]);
            acc[t] = mfma_32x32x2(a[2], b[B0 + 4 * g + 2], acc[t]);
            acc[t] = mfma_32x32x2(a[3], b[B0 + 4 * g + 3], acc[t]);
        }
    }
}

// ---- last chunk of a layer with the layer's epilogue folded in --------------------------------------
// A layer's epilogue (ReLU, mask bits, bias of the next layer: ~5 VALU instructions per accumulator
// register, ~700 per layer) cannot overlap anything when it runs after the last MFMA: one wave per SIMD
// means nobody else feeds the matrix pipe meanwhile (~8 % of a layer).  In the LAST chunk the fragments
// are therefore walked tile-pair-major instead of step-major: output tiles (2P, 2P+1) are final after a
// quarter of the chunk, and their epilogue is issued in eight-register slices between the MFMAs of the
// following pairs; only the last pair's epilogue remains exposed.  (Safe w.r.t. the B operands: the last
// chunk contracts over the registers of the LAST tile only, which the early slices do not write.)
struct NoEpi {
    template <int P, int G, int J> __device__ __forceinline__ void slice() {}
};

template <class EPI, int P, int G, int K>
struct EpiTail {       // slices (P, g, j) for K = 4 g + j = K .. 4 G - 1
    static __device__ __forceinline__ void run(EPI& epi) {
        epi.template slice<P, K / 4, K % 4>();
        if constexpr (K + 1 < 4 * G) EpiTail<EPI, P, G, K + 1>::run(epi);
    }
};

template <int NSTEP, int NT, int CS, int B0, int N_F4, bool CONT_IN, class EPI, int Q, int T0 = 0, int T1 = 0>
struct LastChunk {
    static constexpr int G = CS / 4, NPAIR = NT / 2, NQ = NPAIR * G;
    static constexpr int P = Q / G, g = Q % G;
    static constexpr bool SPREAD = Spread<NQ, N_F4>::kOn;
    static_assert(!SPREAD || Spread<NQ, N_F4>::commit_at(N_F4 - 1) < (NQ * 3) / 4, "LDS writes before the barrier");
    static __device__ __forceinline__ void run(const float (&b)[NSTEP], f32x16 (&acc)[NT], WStream& ws,
                                               f32x4 (&stage)[N_F4 > 0 ? N_F4 : 1], f32x4 (&ring)[8],
                                               const f32x4* A, EPI& epi, float* save_tile) {
        spread_bundle<NQ, N_F4, NSTEP, T0, T1>(Q, ws, stage, b, save_tile);
        if constexpr (!SPREAD && Q == NQ / 2) stream_commit<N_F4>(ws, stage);
        if constexpr (Q == (NQ * 3) / 4) {
            block_sync();
        }
        if constexpr (Q + 1 < NQ) {
            constexpr int P1 = (Q + 1) / G, g1 = (Q + 1) % G;
            ring[2 * ((Q + 1) & 1) + 0] = A[((2 * P1) * G + g1) * 64];
            ring[2 * ((Q + 1) & 1) + 1] = A[((2 * P1 + 1) * G + g1) * 64];
        }
        sched_fence();
        const f32x4 a0 = ring[2 * (Q & 1)], a1 = ring[2 * (Q & 1) + 1];
        constexpr int t = 2 * P;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[t] = mfma_32x32x2(a0[j], b[B0 + 4 * g + j], acc[t]);
            acc[t + 1] = mfma_32x32x2(a1[j], b[B0 + 4 * g + j], acc[t + 1]);
        }
        if constexpr (P >= 1) {
            // the slice is independent of this step's MFMAs: the hints ask for it to be spread between them
            epi.template slice<P - 1, g, 0>();
            epi.template slice<P - 1, g, 1>();
            epi.template slice<P - 1, g, 2>();
            epi.template slice<P - 1, g, 3>();
            if constexpr (lab::kGroupHints) {     // (experiment: explicit MFMA/VALU interleave hints were slower than the default schedule)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    sched_group_mfma<1>();
                    sched_group_valu<EPI::kValuPerMfma>();
                }
            }
        }
        if constexpr (Q + 1 < NQ) {
            LastChunk<NSTEP, NT, CS, B0, N_F4, CONT_IN, EPI, Q + 1, T0, T1>::run(b, acc, ws, stage, ring, A, epi, save_tile);
        } else {
            if constexpr (SPREAD) ws.g += N_F4 * kThreads;
            sched_fence();
            EpiTail<EPI, NPAIR - 1, G, 0>::run(epi);      // the last pair's epilogue: nothing left to hide it under
        }
    }
};

template <int NSTEP, int NT, int CS, int B0, int N_F4, bool CONT_IN, class EPI, int T0 = 0, int T1 = 0>
__device__ __forceinline__ void mfma_chunk_last(const float (&b)[NSTEP], f32x16 (&acc)[NT], WStream& ws,
                                                f32x4 (&stage)[N_F4 > 0 ? N_F4 : 1], f32x4 (&ring)[8],
                                                int lane, EPI& epi, float* save_tile = nullptr) {
    static_assert(NT % 2 == 0 && CS == 16, "tile-pair-major last chunk: 4 step groups");
    constexpr int G = CS / 4;
    constexpr int NQ = (NT / 2) * G;
    if constexpr (!Spread<NQ, N_F4>::kOn) {
        stream_issue<N_F4>(ws, stage);
        sched_fence();
    }
    if constexpr (!Spread<NQ, 1>::kOn) store_tiles<T0, T1, NSTEP>(b, save_tile);
    const f32x4* A = reinterpret_cast<const f32x4*>(WStream::buf(ws.cur)) + lane;
    if constexpr (!CONT_IN) {
        ring[0] = A[(0 * G + 0) * 64];
        ring[1] = A[(1 * G + 0) * 64];
    }
    LastChunk<NSTEP, NT, CS, B0, N_F4, CONT_IN, EPI, 0, T0, T1>::run(b, acc, ws, stage, ring, A, epi, save_tile);
}

template <int NSTEP, int NT, int CS, int NEXT_F4, int C, class EPI = NoEpi>
struct PartLoop {
    static constexpr int NC = NSTEP / CS;
    static constexpr int CHUNK_F4 = NT * CS * 64 / 4 / kThreads;
    static constexpr int N_F4 = (C + 1 < NC) ? CHUNK_F4 : NEXT_F4;
    static constexpr bool CHAIN = !lab::kNoChain && (NT % 2 == 0) && (((CS / 4) * NT / ((NT % 4 == 0) ? kInterleave : 2)) % 2 == 0);
    static __device__ __forceinline__ void run(const float (&b)[NSTEP], f32x16 (&acc)[NT], WStream& ws,
                                               f32x4 (&ring)[8], int lane, float* save_tile, EPI& epi) {
        f32x4 stage[N_F4 > 0 ? N_F4 : 1];
        // the B operands of this part are the previous layer's activations (forward) / this layer's
        // output gradient (backward): the slice this chunk contracts over (tiles T0 .. T1) is stored by the
        // chunk itself, so the training-mode HBM writes trickle out under the MFMAs instead of bursting at a
        // layer end; the chunk also issues the loads and LDS writes of the chunk that follows (Spread)
        constexpr int T0 = CS >= 16 ? (C * CS) / 16 : 0, T1 = CS >= 16 ? ((C + 1) * CS) / 16 : 0;
        if constexpr (C + 1 == NC && !std::is_same<EPI, NoEpi>::value)
            mfma_chunk_last<NSTEP, NT, CS, C * CS, N_F4, CHAIN && (C > 0), EPI, T0, T1>(b, acc, ws, stage, ring, lane,
                                                                                       epi, save_tile);
        else
            mfma_chunk<NSTEP, NT, CS, C * CS, N_F4, CHAIN && (C > 0), CHAIN && (C + 1 < NC), T0, T1>(
                b, acc, ws, stage, ring, lane, save_tile);
        ws.cur = ws.next();
        if constexpr (C + 1 < NC) PartLoop<NSTEP, NT, CS, NEXT_F4, C + 1, EPI>::run(b, acc, ws, ring, lane, save_tile, epi);
    }
};

// One "part" = NSTEP MFMA steps over NT output tiles with B operands taken from registers
// b[0..NSTEP).  CS steps per LDS chunk; NEXT_F4 = 16-byte loads per thread of the chunk that
// follows this part in the stream (0 at the end of the stream).  save_tile: this lane's slot in the
// tile-native HBM section its B operands are saved to (tile_ptr), or nullptr.  The overload with `epi`
// folds the layer's epilogue into the last chunk (see LastChunk).
template <int NSTEP, int NT, int CS, int NEXT_F4>
__device__ __forceinline__ void mfma_part(const float (&b)[NSTEP], f32x16 (&acc)[NT], WStream& ws,
                                          float* save_tile = nullptr) {
    static_assert(NSTEP % CS == 0 && CS % 4 == 0, "chunking");
    static_assert((NT * CS * 64) % (4 * kThreads) == 0, "chunk must be whole 16-byte loads per thread");
    f32x4 ring[8];
    NoEpi none;
    PartLoop<NSTEP, NT, CS, NEXT_F4, 0, NoEpi>::run(b, acc, ws, ring, lane_id(), save_tile, none);
}

template <int NSTEP, int NT, int CS, int NEXT_F4, class EPI>
__device__ __forceinline__ void mfma_part_epi(const float (&b)[NSTEP], f32x16 (&acc)[NT], WStream& ws,
                                              float* save_tile, EPI& epi) {
    static_assert(NSTEP % CS == 0 && CS % 4 == 0, "chunking");
    f32x4 ring[8];
    PartLoop<NSTEP, NT, CS, NEXT_F4, 0, EPI>::run(b, acc, ws, ring, lane_id(), save_tile, epi);
}

// acc[t][r] = bias of feature feat_of(t, r, h); `table` is in lane-vector layout (mlp_layout.
// lane_vector_table): entry ((4 t + q) * 2 + h) * 4 + j  <->  register 4 q + j of tile t on lane half h.
__device__ __forceinline__ f32x4 lane_vec(const float* __restrict__ table, int t, int q, int h) {
    return *reinterpret_cast<const f32x4*>(table + ((4 * t + q) * 2 + h) * 4);
}

template <int NT>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NT], const float* __restrict__ table, int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = lane_vec(table, t, q, h);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][4 * q + j] = b[j];
        }
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// this lane's slot in the tile-native section of `width` features that starts at `section`.  The result is
// declared non-null: the stores behind it test the pointer (nullptr = nothing to save, the inference
// instantiation), and a per-lane test the compiler cannot fold is a divergent branch around EVERY store --
// 32 extra basic blocks per trunk layer, each a barrier for the instruction scheduler.
__device__ __forceinline__ float* tile_ptr(float* section, long wave_tile, int width, int lane) {
    float* p = section + wave_tile * (32L * width) + lane * 4;
    __builtin_assume(p != nullptr);
    return p;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline long padded_samples(long P) {
    return (P + kSamplesPerBlock - 1) / kSamplesPerBlock * kSamplesPerBlock;
}

// ReLU mask of N <= 128 registers in 4 words: element i sits in word i >> 5 at bit 31 - (i & 31)
// (built by shifting in from the right: bits = 2 * bits + (v > 0) is one v_cmp + one v_addc_co).
template <int N>
__device__ __forceinline__ u32x4 relu_bits(const float (&v)[N]) {
    u32x4 bits = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < N; ++i) bits[i >> 5] = shift_in_positive(bits[i >> 5], v[i]);
    return bits;
}

// v if the mask bit of element i is set, else +0 (v_bfe_i32 + v_and_b32; i must be a compile-time constant
// after unrolling)
__device__ __forceinline__ float mask_select(float v, u32x4 bits, int i) {
    const unsigned w = bits[i >> 5];
    switch (31 - (i & 31)) {
#define SCN_CASE(P) case P: return keep_if_bit<P>(v, w);
        SCN_CASE(0) SCN_CASE(1) SCN_CASE(2) SCN_CASE(3) SCN_CASE(4) SCN_CASE(5) SCN_CASE(6) SCN_CASE(7)
        SCN_CASE(8) SCN_CASE(9) SCN_CASE(10) SCN_CASE(11) SCN_CASE(12) SCN_CASE(13) SCN_CASE(14) SCN_CASE(15)
        SCN_CASE(16) SCN_CASE(17) SCN_CASE(18) SCN_CASE(19) SCN_CASE(20) SCN_CASE(21) SCN_CASE(22) SCN_CASE(23)
        SCN_CASE(24) SCN_CASE(25) SCN_CASE(26) SCN_CASE(27) SCN_CASE(28) SCN_CASE(29) SCN_CASE(30) SCN_CASE(31)
#undef SCN_CASE
    }
    return v;
}

template <int PD>
__device__ __forceinline__ unsigned int* mask_ptr(float* save, long P, int section, long wave_tile, int lane) {
    unsigned int* base = reinterpret_cast<unsigned int*>(save + (long)Var<PD>::kSavePerSample * padded_samples(P));
    return base + ((long)section * (padded_samples(P) / 32) + wave_tile) * 256 + lane * 4;
}

}  // namespace mlp
}  // namespace scn
