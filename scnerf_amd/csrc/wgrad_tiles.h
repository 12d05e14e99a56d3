// wgrad_tiles.h -- the narrower weight-gradient GEMMs of a network pass, same scheme as wgrad256.h:
//
//     dW[n][k] = sum_p dZ[p][n] * X[p][k]      n < NA, k < NB,      db[n] = sum_p dZ[p][n]
//
//   NA x NB        A (dZ)                    B (X)                          reference layer (run_nerf_helpers.py:92-128)
//   256 x  64      trunk dZ, tile-native     encoded points, row-major      pts_linears[0], skip part of pts_linears[5]
//   256 x 128      trunk dZ, tile-native     encoded 4-D points, row-major  the same layers of NeRF++'s background net
//   128 x 256      views-layer dZ, tile-nat. feature, tile-native           views_linears[0] (feature part)
//
// A workgroup (2 x 2 waves) owns the whole NA x NB output for a chunk of samples.  Wave (wn, wk) covers NA/2
// rows and NB/2 columns as TA x TB accumulator tiles with TA = NA/64, TB = NB/64: lane l & 31 fetches TA
// (TB) CONSECUTIVE floats of a sample's row with one ds_read_b128 / b64 / b32 -- its operands of TA (TB)
// different tiles for the same k-step, tile row i standing for feature TA i + a (wgrad256.h explains the
// mapping).  The previous kernel read every operand with ds_read_b32: at one wave per SIMD that rate (a fifth
// of the LDS peak) was what bounded these narrow shapes, 5 reads for 4 MFMAs on the 256 x 64 one.
// Loads and LDS writes are spread over the k-steps; bias sums ride on the staged dZ pieces.  A stage of these
// shapes is only 1.7-3.4 us of MFMA work -- about one HBM round trip under load --, so the loads run TWO
// stages ahead through two sets of staging registers: stage st + 2 is fetched while st computes and written
// to LDS while st + 1 computes.
//
// A row-major B operand ([P][NB], ld = NB) may hold anything in rows >= P (the forward only writes valid
// samples): those rows are staged as zeros.  dZ is exactly zero there (mlp_bwd.hip feeds zero d_raw).
#pragma once
#include <type_traits>

#include <scn_wave.h>

namespace scn {
namespace wgt {

constexpr int kThreads = 256;
constexpr int kMS = 32;
constexpr int kSteps = kMS / 2;

struct Args {
    const float* A;       // dZ, tile-native, width NA
    const float* B;       // X: tile-native width NB, or row-major [P][NB]
    float* part_w;        // [G][NA][NB]
    float* part_b;        // [G][NA] or nullptr
    long P;               // valid samples
    long Ppad;            // samples the tile-native sections cover
    long chunk;           // samples per workgroup (multiple of kMS)
};

template <int N>
struct Vec;
template <> struct Vec<1> { typedef float type; };
template <> struct Vec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec<4> { typedef f32x4 type; };

template <int N>
__device__ __forceinline__ void read_vec(const float* p, float (&out)[N]) {
    if constexpr (N == 1) {
        out[0] = *p;
    } else {
        const typename Vec<N>::type v = *reinterpret_cast<const typename Vec<N>::type*>(p);
#pragma unroll
        for (int i = 0; i < N; ++i) out[i] = v[i];
    }
}

template <int NA, int NB, bool B_ROWMAJOR>
constexpr unsigned lds_bytes() { return 2u * kMS * (NA + 4 + NB + 4) * sizeof(float); }

// timing-experiment switches (tools/ubench/tiles_lab.hip); the product instantiates FLAGS = 0
enum : int { kNoLoad = 1, kNoBarrier = 2, kNoBias = 4 };

template <int NA, int NB, bool B_ROWMAJOR, int FLAGS = 0>
__global__ __launch_bounds__(kThreads, 1) void wgrad_tiles_kernel(Args a) {
    constexpr int TA = NA / 64, TB = NB / 64;
    static_assert((TA == 2 || TA == 4) && (TB == 1 || TB == 2 || TB == 4), "wave tile shapes");
    constexpr int LDA = NA + 4, LDB = NB + 4;
    constexpr int OPA = kMS * LDA, STAGE = kMS * (LDA + LDB);
    constexpr int PA = kMS * NA / 4 / kThreads, PB = kMS * NB / 4 / kThreads;    // 16-byte pieces per thread
    constexpr int NP = PA + PB;
    float* lds = dynamic_lds<float>();
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const int wn = wave >> 1, wk = wave & 1;
    const int li = lane & 31, mh = lane >> 5;
    // 32-sample tiles are dealt to the workgroups round-robin (tile T -> workgroup T % G): at any moment the
    // grid streams one contiguous region of each operand, spread over every HBM channel, instead of G streams
    // a fixed power-of-two-ish stride apart
    const long n_tiles = a.Ppad / kMS;
    const int G = gridDim.x;
    const int n_stage = (long)blockIdx.x < n_tiles ? (int)((n_tiles - 1 - blockIdx.x) / G) + 1 : 0;
    float* const pw_block = a.part_w + (long)blockIdx.x * NA * NB;
    float* const pb_block = a.part_b ? a.part_b + (long)blockIdx.x * NA : nullptr;

    if (n_stage == 0) {
        for (int e = tid * 4; e < NA * NB; e += kThreads * 4)
            *reinterpret_cast<f32x4*>(pw_block + e) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (pb_block && tid < NA) pb_block[tid] = 0.f;
        return;
    }

    f32x16 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 bsum[PA];
#pragma unroll
    for (int q = 0; q < PA; ++q) bsum[q] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staged pieces.  Tile-native operand of width NX: piece q * 256 + tid = (t * 4 + qq) * 64 + lane with
    // t = q, qq = wave -> sample li, columns 32 q + 8 wave + 4 mh .. +3, source offset (q * 256 + tid) * 4.
    // Row-major operand [32][NB]: piece e = q * 256 + tid -> row 4 e / NB, columns 4 e % NB .. +3.
    const int dst_a = li * LDA + wave * 8 + mh * 4;               // + q * 32
    int dst_b[PB], row_b[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int e = (q * kThreads + tid) * 4;
        if constexpr (B_ROWMAJOR) { row_b[q] = e / NB; dst_b[q] = OPA + (e / NB) * LDB + e % NB; }
        else { row_b[q] = 0; dst_b[q] = OPA + li * LDB + q * 32 + wave * 8 + mh * 4; }
    }
    const int rd_a = mh * LDA + (NA / 2) * wn + TA * li;          // + s * 2 LDA
    const int rd_b = OPA + mh * LDB + (NB / 2) * wk + TB * li;    // + s * 2 LDB

    f32x4 sa[2][PA], sb[2][PB];
    const float* nextA = nullptr;
    const float* nextB = nullptr;
    int rows_valid = kMS;
    auto locate = [&](int st) {
        const long p0 = ((long)st * G + blockIdx.x) * kMS;
        nextA = a.A + p0 * NA + tid * 4;
        nextB = a.B + p0 * NB + tid * 4;
        rows_valid = (int)min((long)kMS, a.P - p0);
    };
    auto sync = [&]() {
        if constexpr (!(FLAGS & kNoBarrier)) block_sync();
    };
    auto load_piece = [&](int set, int i) {
        if constexpr (FLAGS & kNoLoad) return;
        if (i < PA) {
            sa[set][i] = *reinterpret_cast<const f32x4*>(nextA + i * (kThreads * 4));
        } else {
            const int q = i - PA;
            if constexpr (B_ROWMAJOR) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row_b[q] < rows_valid) v = *reinterpret_cast<const f32x4*>(nextB + q * (kThreads * 4));
                sb[set][q] = v;
            } else {
                sb[set][q] = *reinterpret_cast<const f32x4*>(nextB + q * (kThreads * 4));
            }
        }
    };
    auto commit_piece = [&](int buf, int set, int i) {
        if constexpr (FLAGS & kNoLoad) return;
        float* s = lds + buf * STAGE;
        if (i < PA) {
            *reinterpret_cast<f32x4*>(s + dst_a + i * 32) = sa[set][i];
            // bias sums: every staged dZ piece is added exactly once, when it goes to LDS
            if constexpr (!(FLAGS & kNoBias)) {
                bsum[i][0] = add_raw(bsum[i][0], sa[set][i][0]); bsum[i][1] = add_raw(bsum[i][1], sa[set][i][1]);
                bsum[i][2] = add_raw(bsum[i][2], sa[set][i][2]); bsum[i][3] = add_raw(bsum[i][3], sa[set][i][3]);
            }
        } else {
            *reinterpret_cast<f32x4*>(s + dst_b[i - PA]) = sb[set][i - PA];
        }
    };
    constexpr int kLoadSteps = kSteps / 4, kCommitFrom = kSteps / 2;
    // Stage st (LDS buffer `buf`): FETCH loads stage st + 2 into register set LS = st & 1 (free: its pieces went
    // to LDS during stage st - 1); COMMIT writes set 1 - LS (stage st + 1) to the other LDS buffer.
    auto stage = [&](int buf, auto ls_tag, auto fetch_tag, auto commit_tag, int fetch_st) {
        constexpr int LS = decltype(ls_tag)::value;
        constexpr bool FETCH = decltype(fetch_tag)::value, COMMIT = decltype(commit_tag)::value;
        if constexpr (FETCH) locate(fetch_st);
        const float* As = lds + buf * STAGE + rd_a;
        const float* Bs = lds + buf * STAGE + rd_b;
        float av[2][TA], bv[2][TB];
        read_vec<TA>(As, av[0]);
        read_vec<TB>(Bs, bv[0]);
#pragma unroll
        for (int s = 0; s < kSteps; ++s) {
            const int cur = s & 1;
            if (s + 1 < kSteps) {
                read_vec<TA>(As + (s + 1) * 2 * LDA, av[cur ^ 1]);
                read_vec<TB>(Bs + (s + 1) * 2 * LDB, bv[cur ^ 1]);
            }
            if constexpr (FETCH) {
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    if ((i * kLoadSteps) / NP == s) load_piece(LS, i);
            }
            if constexpr (COMMIT) {
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    if (kCommitFrom + (i * (kSteps - kCommitFrom)) / NP == s) commit_piece(buf ^ 1, 1 - LS, i);
            }
            sched_fence();
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j) acc[i][j] = mfma_32x32x2(av[cur][i], bv[cur][j], acc[i][j]);
        }
        sync();
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using Y = std::true_type;
    using N = std::false_type;

    locate(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) load_piece(0, i);
#pragma unroll
    for (int i = 0; i < NP; ++i) commit_piece(0, 0, i);
    if (n_stage > 1) {
        locate(1);
#pragma unroll
        for (int i = 0; i < NP; ++i) load_piece(1, i);
    }
    block_sync();
    int st = 0;
    for (; st + 3 < n_stage; st += 2) {                  // both stages of the pair still fetch: no conditions inside
        stage(0, S0{}, Y{}, Y{}, st + 2);
        stage(1, S1{}, Y{}, Y{}, st + 3);
    }
    const int rest = n_stage - st;                       // 1, 2 or 3 stages left (st is even: buffer 0, set 0)
    if (rest == 3) {
        stage(0, S0{}, Y{}, Y{}, st + 2);
        stage(1, S1{}, N{}, Y{}, 0);
        stage(0, S0{}, N{}, N{}, 0);
    } else if (rest == 2) {
        stage(0, S0{}, N{}, Y{}, 0);
        stage(1, S1{}, N{}, N{}, 0);
    } else {
        stage(0, S0{}, N{}, N{}, 0);
    }

    // ---- partial sums: tile (i, j) element r of lane (li, mh) is
    //      dW[(NA/2) wn + TA (r&3 + 8 (r>>2) + 4 mh) + i][(NB/2) wk + TB li + j]
    float* pw = pw_block + (NB / 2) * wk + TB * li;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = (NA / 2) * wn + TA * ((r & 3) + 8 * (r >> 2) + 4 * mh) + i;
            if constexpr (TB == 1) {
                pw[n * NB] = acc[i][0][r];
            } else {
                typename Vec<TB>::type v;
#pragma unroll
                for (int j = 0; j < TB; ++j) v[j] = acc[i][j][r];
                *reinterpret_cast<typename Vec<TB>::type*>(pw + n * NB) = v;
            }
        }
    // bias: fold over the 32 in-tile samples (lanes of one half); piece q covers columns 32 q + 8 wave + 4 mh .. +3
#pragma unroll
    for (int q = 0; q < PA; ++q) {
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

}  // namespace wgt
}  // namespace scn
