// wgrad256.h -- the 256 x 256 weight-gradient GEMMs of one network pass as ONE launch.
//
//     dW_j[n][k] = sum_p dZ_j[p][n] * X_j[p][k]      (j = 0 .. n_jobs-1; n, k < 256)
//     db_j[n]    = sum_p dZ_j[p][n]
//
// What autograd derives for the 256-wide nn.Linear layers of the reference network
// (/root/reference NeRF/run_nerf_helpers.py:92-103, :105-128): trunk layers 1-7 (the hidden part of the
// skip layer included) and feature_linear -- 8 of the 12 GEMMs and 87 % of the weight-gradient FLOPs.
//
// Both operands are tile-native sections of width 256 (mlp_common.h): per 32-sample tile a block
// [t][q][lane][4] whose 16-byte piece (t, q, lane = m + 32 h) holds features 32 t + 8 q + 4 h .. +3 of
// sample m.  A workgroup (2 x 2 waves, one per SIMD) owns the whole 256 x 256 output of ONE job for a
// contiguous chunk of samples; the grid is (chunks, jobs), so the workgroups of job j + 1 are dispatched
// onto CUs as those of job j retire: one launch, the GEMMs' tails overlap instead of adding up.
//
// Operand mapping.  v_mfma_f32_32x32x2_f32 takes ONE float per lane and operand, A[i = l & 31][k = l >> 5].
// Which feature a tile row stands for is free, so row i of accumulator tile (a, b) is feature 4 i + a of
// the wave's 128 and column j is input 4 j + b: the four consecutive floats a lane fetches with one
// ds_read_b128 from the row-major LDS image [32 samples][256 + 4] are its A operands of four different
// tiles for the same k-step.  Per k-step (2 samples: lanes < 32 read sample 2 s, lanes >= 32 sample
// 2 s + 1) a wave issues 2 x ds_read_b128 for 16 MFMAs (the previous kernel: 8 x ds_read_b32); 32 lanes of
// a half read 512 contiguous bytes, so the reads are conflict-free whatever the row stride; the +4 pad
// keeps the 16-byte de-tiling writes conflict-free (8 consecutive samples x 4 floats = 32 banks).
//
// Bias gradients ride on the staging registers: thread `tid` always stages the same 8 column groups of
// the same in-tile sample, so it adds every staged dZ piece into 32 private sums (32 VALU adds per stage
// instead of 64 on the MFMA operands); they are folded over the 32 samples with shuffles once per job.
#pragma once
#include <type_traits>

#include <scn_wave.h>

namespace scn {
namespace wg256 {

constexpr int kThreads = 256;
constexpr int kMS = 32;                       // samples per stage = one wave tile of the MLP kernels
constexpr int kW = 256;
constexpr int kLd = kW + 4;                   // padded LDS row stride (floats)
constexpr int kOperand = kMS * kLd;           // floats of one staged operand
constexpr int kStage = 2 * kOperand;          // A then B
constexpr int kPieces = kMS * kW / 4 / kThreads;   // 16-byte pieces per thread, stage and operand (8)
constexpr int kSteps = kMS / 2;               // k-steps (MFMA contractions of 2 samples) per stage
constexpr int kMaxJobs = 10;
constexpr unsigned kLdsBytes = 2u * kStage * sizeof(float);     // 133 120: one workgroup per CU
static_assert(kPieces == 8 && kSteps == 16, "stage shape");

struct Job {
    const float* A;       // dZ, tile-native width 256
    const float* B;       // X,  tile-native width 256
    float* part_w;        // [G][256][256]
    float* part_b;        // [G][256] or nullptr
};
struct Args {
    Job job[kMaxJobs];
    int n_jobs;
    long Ppad;            // samples the tile-native sections cover (multiple of 128)
    long chunk;           // samples per workgroup (multiple of kMS)
};

// timing-experiment switches (tools/ubench/wgrad_lab.hip); the product instantiates FLAGS = kSpread
enum : int {
    kNoLoad = 1,          // no global loads / LDS commits (reads whatever the LDS holds)
    kNoBarrier = 2,       // no workgroup barriers (results are wrong)
    kNoBias = 4,          // no bias sums
    kSpread = 8,          // loads and commits spread over the k-steps instead of two bursts per stage
};

template <int FLAGS>
__global__ __launch_bounds__(kThreads, 1) void wgrad256_kernel(Args a) {
    float* lds = dynamic_lds<float>();
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const int wn = wave >> 1, wk = wave & 1;
    const int li = lane & 31, mh = lane >> 5;
    const Job& J = a.job[blockIdx.y];
    const long p_begin = (long)blockIdx.x * a.chunk;
    const long p_end = min(a.Ppad, p_begin + a.chunk);
    const int n_stage = p_begin < p_end ? (int)((p_end - p_begin + kMS - 1) / kMS) : 0;
    float* const pw_block = J.part_w + (long)blockIdx.x * kW * kW;
    float* const pb_block = J.part_b ? J.part_b + (long)blockIdx.x * kW : nullptr;

    if (n_stage == 0) {
        // more workgroups than sample tiles: this chunk is empty, its slab must still read as zero
        for (int e = tid * 4; e < kW * kW; e += kThreads * 4)
            *reinterpret_cast<f32x4*>(pw_block + e) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (pb_block && tid < kW) pb_block[tid] = 0.f;
        return;
    }

    f32x16 acc[4][4];
    f32x4 bsum[kPieces];
    auto clear = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int q = 0; q < kPieces; ++q) bsum[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    clear();

    // this thread's staged piece q: tile-native piece index q * 256 + tid = (t * 4 + qq) * 64 + lane with
    // t * 4 + qq = 4 q + wave -> sample li, columns 32 q + 8 wave + 4 mh .. +3
    const unsigned src0 = tid * 4;                                // + q * 1024 floats
    const int dst0 = li * kLd + wave * 8 + mh * 4;                // + q * 32 floats
    const int rd_a = mh * kLd + 128 * wn + 4 * li;                // + s * 2 kLd
    const int rd_b = kOperand + mh * kLd + 128 * wk + 4 * li;

    f32x4 sa[kPieces], sb[kPieces];
    // wave-uniform block bases of the stage being loaded (SGPR pairs) + one 32-bit per-thread byte offset:
    // the loads are `global_load_dwordx4 v, v_off, s[base]`, no 64-bit vector address arithmetic
    const char* nextA = nullptr;
    const char* nextB = nullptr;
    const unsigned voff = src0 * (unsigned)sizeof(float);
    auto locate = [&](int st) {
        const long p0 = p_begin + (long)st * kMS;
        nextA = reinterpret_cast<const char*>(J.A + p0 * kW);
        nextB = reinterpret_cast<const char*>(J.B + p0 * kW);
    };
    auto load_piece = [&](int q) {
        if constexpr (!(FLAGS & kNoLoad)) {
            sa[q] = *reinterpret_cast<const f32x4*>(nextA + (size_t)q * (kThreads * 16) + voff);
            sb[q] = *reinterpret_cast<const f32x4*>(nextB + (size_t)q * (kThreads * 16) + voff);
        }
    };
    auto commit_piece = [&](int buf, int q) {
        if constexpr (!(FLAGS & kNoLoad)) {
            float* s = lds + buf * kStage + dst0 + q * 32;
            *reinterpret_cast<f32x4*>(s) = sa[q];
            *reinterpret_cast<f32x4*>(s + kOperand) = sb[q];
        }
    };
    // bias sums: the staged dZ piece of the CURRENT stage is added while it is still in its staging
    // register, i.e. before the load of the next stage's piece overwrites it (so a piece is always
    // accounted to the job whose MFMAs consume it)
    auto account_piece = [&](int q) {
        if constexpr (!(FLAGS & (kNoBias | kNoLoad))) {
            bsum[q][0] = add_raw(bsum[q][0], sa[q][0]); bsum[q][1] = add_raw(bsum[q][1], sa[q][1]);
            bsum[q][2] = add_raw(bsum[q][2], sa[q][2]); bsum[q][3] = add_raw(bsum[q][3], sa[q][3]);
        }
    };
    auto sync = [&]() {
        if constexpr (!(FLAGS & kNoBarrier)) block_sync();
    };

    // one stage = 32 samples out of LDS buffer `buf`; MORE: another stage follows, whose operands are loaded
    // and committed to the other buffer under this stage's MFMAs
    auto stage = [&](int buf, auto more_tag, int next_st) {
        constexpr bool MORE = decltype(more_tag)::value;
        if constexpr (MORE) locate(next_st);
        const float* As = lds + buf * kStage + rd_a;
        const float* Bs = lds + buf * kStage + rd_b;
        if constexpr (!(FLAGS & kSpread)) {
#pragma unroll
            for (int q = 0; q < kPieces; ++q) account_piece(q);
            if constexpr (MORE) {
#pragma unroll
                for (int q = 0; q < kPieces; ++q) load_piece(q);
            }
            sched_fence();
        }
        f32x4 av[2], bv[2];
        av[0] = *reinterpret_cast<const f32x4*>(As);
        bv[0] = *reinterpret_cast<const f32x4*>(Bs);
#pragma unroll
        for (int s = 0; s < kSteps; ++s) {
            const int cur = s & 1;
            if (s + 1 < kSteps) {
                av[cur ^ 1] = *reinterpret_cast<const f32x4*>(As + (s + 1) * 2 * kLd);
                bv[cur ^ 1] = *reinterpret_cast<const f32x4*>(Bs + (s + 1) * 2 * kLd);
            }
            if constexpr (FLAGS & kSpread) {
                // next stage: two loads per step over the first four steps, one commit per step over the last
                // eight, so that no burst of non-MFMA instructions stalls the matrix pipe
                if (s < 4) {
                    account_piece(2 * s);
                    account_piece(2 * s + 1);
                    if constexpr (MORE) { load_piece(2 * s); load_piece(2 * s + 1); }
                }
                if constexpr (MORE) {
                    if (s >= kSteps - kPieces) commit_piece(buf ^ 1, s - (kSteps - kPieces));
                }
            } else if constexpr (MORE) {
                if (s == kSteps * 3 / 4) {
#pragma unroll
                    for (int q = 0; q < kPieces; ++q) commit_piece(buf ^ 1, q);
                }
            }
            sched_fence();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma_32x32x2(av[cur][i], bv[cur][j], acc[i][j]);
        }
        sync();
    };

    locate(0);
#pragma unroll
    for (int q = 0; q < kPieces; ++q) load_piece(q);
#pragma unroll
    for (int q = 0; q < kPieces; ++q) commit_piece(0, q);
    sync();

    int buf = 0;
    for (int st = 0; st + 1 < n_stage; ++st) {
        stage(buf, std::true_type{}, st + 1);
        buf ^= 1;
    }
    stage(buf, std::false_type{}, 0);

    // ---- partial sums: tile (i, j) element r of lane (li, mh) is
    //      dW[128 wn + 4 (r&3 + 8 (r>>2) + 4 mh) + i][128 wk + 4 li + j]
    float* pw = pw_block + 128 * wk + 4 * li;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = 128 * wn + 4 * ((r & 3) + 8 * (r >> 2) + 4 * mh) + i;
            const f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            *reinterpret_cast<f32x4*>(pw + n * kW) = v;
        }
    if constexpr (!(FLAGS & kNoBias)) {
        // fold over the 32 in-tile samples (lanes of one half); piece q covers columns 32 q + 8 wave + 4 mh .. +3
#pragma unroll
        for (int q = 0; q < kPieces; ++q) {
            f32x4 v = bsum[q];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float x = v[c];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) x += shfl_xor(x, o);
                v[c] = x;
            }
            if (pb_block && li == 0) *reinterpret_cast<f32x4*>(pb_block + 32 * q + 8 * wave + 4 * mh) = v;
        }
    }
}

}  // namespace wg256
}  // namespace scn
