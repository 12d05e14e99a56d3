// wgrad.hip -- weight / bias gradients of one linear layer as an fp32-MFMA GEMM reduced over the
// samples:  dW[n][k] = sum_p dZ[p][n] * X[p][k],  db[n] = sum_p dZ[p][n]   (+ an optional rank-1
// side product  dv[k] = sum_p v[p] * X[p][k]  used for alpha_linear, whose single output row
// would waste a whole MFMA tile).
//
// What autograd derives for every nn.Linear of the reference network
// (/root/reference NeRF/run_nerf_helpers.py:13-21, :105-128).
//
// dZ and X are the row-major [P][ld] tensors left in HBM by mlp_bwd.hip / the training forward,
// so both MFMA operands are 128-byte row segments: lane l supplies A[i = l&31][k = l>>5] =
// dZ[p = 2s + (l>>5)][n0 + (l&31)] and B likewise from X.  A workgroup (4 waves, 2 x 2) owns the
// whole [BN x BK] output for a contiguous chunk of samples, staging 16 samples at a time through
// double-buffered LDS; partial results go to a workspace and a second kernel reduces them in a
// fixed order (deterministic; 64 MB per 256x256 layer at 256 chunks is ~25 us of HBM time).
#include <cstdlib>
#include <type_traits>

#include <scn_wave.h>

#include "launch.h"
#include "mlp_common.h"
#include "mlp_h3.h"
#include "scnerf_hip.h"
#include "wgrad256.h"
#include "wgrad256_half.h"
#include "wgrad_half_narrow.h"
#include "wgrad_tiles.h"

namespace {

using namespace scn;

constexpr int kThreads = 256;
constexpr int kMS = 32;   // samples per LDS stage = one wave tile of the MLP kernels

struct WgradArgs {
    const float* A; int lda; int n_load; int a_tiled;   // dZ: tile-native section of width lda, or row-major [P][lda]
    const float* B; int ldb; int k_load; int b_tiled;   // X : likewise
    long P;                                             // valid samples (row-major operands are masked beyond)
    long Ppad;                                          // samples the tile-native sections cover
    long chunk;                                         // samples per workgroup (multiple of kMS)
    float* part_w;                                      // [G][BN][BK]
    float* part_b;                                      // [G][BN]
};

// LDS image of one staged operand: row-major [kMS samples][width + 4] (the 4-float pad makes the
// 16-byte writes of 8 consecutive samples and the 4-byte reads of 32 consecutive columns both
// conflict-free).  A 16-byte piece of either HBM layout holds 4 consecutive columns of one sample:
//   row-major source : piece e -> sample e / width, column e % width
//   tile-native source: piece index e/4 = (t*4 + q)*64 + lane, lane = m + 32 h -> sample m,
//                       column 32 t + 8 q + 4 h
__device__ __forceinline__ int lds_piece_offset(int e, int width, int tiled) {
    int m, c;
    if (tiled) {
        const int piece = e >> 2, ln = piece & 63, tq = piece >> 6;
        m = ln & 31;
        c = (tq >> 2) * 32 + (tq & 3) * 8 + (ln >> 5) * 4;
    } else {
        m = e / width;
        c = e % width;
    }
    return m * (width + 4) + c;
}

// FAST: both operands tile-native -- every staged piece exists (the sections are padded), so the loads are
// unconditional; otherwise rows beyond P / columns beyond n_load of a row-major operand read as zero.
template <int WN, int WK, bool FAST>
__global__ __launch_bounds__(kThreads, 1) void wgrad_kernel(WgradArgs a) {
    constexpr int BN = 2 * WN * 32, BK = 2 * WK * 32;
    constexpr int LDA = BN + 4, LDB = BK + 4;              // padded LDS row strides
    constexpr int STAGE = kMS * (LDA + LDB);               // floats per LDS stage
    constexpr int A_F4 = kMS * BN / 4 / kThreads, B_F4 = kMS * BK / 4 / kThreads;
    static_assert(kMS * BN % (4 * kThreads) == 0 && kMS * BK % (4 * kThreads) == 0, "stage shape");
    float* lds = dynamic_lds<float>();
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const int wn = wave >> 1, wk = wave & 1;
    const int wk_uniform = uniform(wk);
    const int li = lane & 31, mh = lane >> 5;
    const long p_begin = (long)blockIdx.x * a.chunk;
    const long p_lim = a.Ppad;                              // tiles exist up to here
    const long p_end = min(p_lim, p_begin + a.chunk);
    const int n_stage = p_begin < p_end ? (int)((p_end - p_begin + kMS - 1) / kMS) : 0;

    f32x16 acc[WN][WK];
#pragma unroll
    for (int i = 0; i < WN; ++i)
#pragma unroll
        for (int j = 0; j < WK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[WN];
#pragma unroll
    for (int i = 0; i < WN; ++i) bsum[i] = 0.f;

    int a_dst[A_F4], b_dst[B_F4];      // where this thread's staged pieces go in the LDS image
#pragma unroll
    for (int q = 0; q < A_F4; ++q) a_dst[q] = lds_piece_offset((q * kThreads + tid) * 4, BN, a.a_tiled);
#pragma unroll
    for (int q = 0; q < B_F4; ++q) b_dst[q] = lds_piece_offset((q * kThreads + tid) * 4, BK, a.b_tiled);

    // per-thread source offsets of the staged 16-byte pieces, relative to the stage's first sample
    // (computed once: the address arithmetic must not sit in front of every stage's MFMAs)
    int a_src[A_F4], b_src[B_F4], a_row[A_F4], b_row[B_F4];
#pragma unroll
    for (int q = 0; q < A_F4; ++q) {
        const int e = (q * kThreads + tid) * 4;
        if (a.a_tiled) { a_src[q] = e; a_row[q] = 0; }
        else { a_src[q] = (e / BN) * a.lda + e % BN; a_row[q] = (e % BN) < a.n_load ? e / BN : kMS; }
    }
#pragma unroll
    for (int q = 0; q < B_F4; ++q) {
        const int e = (q * kThreads + tid) * 4;
        if (a.b_tiled) { b_src[q] = e; b_row[q] = 0; }
        else { b_src[q] = (e / BK) * a.ldb + e % BK; b_row[q] = (e % BK) < a.k_load ? e / BK : kMS; }
    }
    f32x4 sa[A_F4], sb[B_F4];
    auto issue = [&](int st) {
        const long p0 = p_begin + (long)st * kMS;
        // tile-native: the 32-sample block starts at p0 * ld; row-major: row p0.  Rows >= P of a
        // row-major operand are zero (tile-native sections hold zeros / finite values there).
        const float* As0 = a.A + p0 * a.lda;
        const float* Bs0 = a.B + p0 * a.ldb;
        const int rows = (int)min((long)kMS, a.P - p0);       // valid rows of a row-major operand
        const int rows_a = a.a_tiled ? kMS : rows, rows_b = a.b_tiled ? kMS : rows;
        if constexpr (FAST) {
#pragma unroll
            for (int q = 0; q < A_F4; ++q) sa[q] = *reinterpret_cast<const f32x4*>(As0 + (q * kThreads + tid) * 4);
#pragma unroll
            for (int q = 0; q < B_F4; ++q) sb[q] = *reinterpret_cast<const f32x4*>(Bs0 + (q * kThreads + tid) * 4);
        } else {
#pragma unroll
            for (int q = 0; q < A_F4; ++q) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (a_row[q] < rows_a) v = *reinterpret_cast<const f32x4*>(As0 + a_src[q]);
                sa[q] = v;
            }
#pragma unroll
            for (int q = 0; q < B_F4; ++q) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (b_row[q] < rows_b) v = *reinterpret_cast<const f32x4*>(Bs0 + b_src[q]);
                sb[q] = v;
            }
        }
    };
    auto commit = [&](int buf) {
        float* s = lds + buf * STAGE;
#pragma unroll
        for (int q = 0; q < A_F4; ++q) *reinterpret_cast<f32x4*>(s + a_dst[q]) = sa[q];
#pragma unroll
        for (int q = 0; q < B_F4; ++q) *reinterpret_cast<f32x4*>(s + kMS * LDA + b_dst[q]) = sb[q];
    };

    if (n_stage > 0) {
        issue(0);
        commit(0);
    }
    block_sync();
    // (unrolling the stage loop by two to make `buf` a compile-time constant spills: 632 B/lane of scratch)
    for (int st = 0; st < n_stage; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < n_stage;
        if (more) issue(st + 1);
        sched_fence();          // the global loads stay at the head of the stage
        const float* As = lds + buf * STAGE;
        const float* Bs = As + kMS * LDA;
        // operands of step s+1 are read while the WN x WK MFMAs of step s run
        float av[2][WN], bv[2][WK];
        auto read_step = [&](int s2, int slot) {
            const int m = 2 * s2 + mh;
#pragma unroll
            for (int i = 0; i < WN; ++i) av[slot][i] = As[m * LDA + (wn * WN + i) * 32 + li];
#pragma unroll
            for (int j = 0; j < WK; ++j) bv[slot][j] = Bs[m * LDB + (wk * WK + j) * 32 + li];
        };
        read_step(0, 0);
#pragma unroll
        for (int s = 0; s < kMS / 2; ++s) {
            const int cur = s & 1;
            if (s + 1 < kMS / 2) read_step(s + 1, cur ^ 1);
            if (s == (kMS / 2) * 3 / 4 && more) commit(buf ^ 1);
            sched_fence();
#pragma unroll
            for (int i = 0; i < WN; ++i)
#pragma unroll
                for (int j = 0; j < WK; ++j) acc[i][j] = mfma_32x32x2(av[cur][i], bv[cur][j], acc[i][j]);
            if (wk_uniform == 0) {       // wave-uniform: a real branch, not 2 x WN selects for every wave
#pragma unroll
                for (int i = 0; i < WN; ++i) bsum[i] += av[cur][i];
            }
        }
        block_sync();
    }

    // ---- partial results -----------------------------------------------------------------
    float* pw = a.part_w + (long)blockIdx.x * BN * BK;
#pragma unroll
    for (int i = 0; i < WN; ++i)
#pragma unroll
        for (int j = 0; j < WK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = (wn * WN + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * mh;
                const int k = (wk * WK + j) * 32 + li;
                pw[n * BK + k] = acc[i][j][r];
            }
    if (wk == 0) {
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const float tot = bsum[i] + shfl_xor(bsum[i], 32);
            if (mh == 0) a.part_b[(long)blockIdx.x * BN + (wn * WN + i) * 32 + li] = tot;
        }
    }
}

// dv[k] = sum_p v[p] * X[p][k] for a tile-native X of width 256 (alpha_linear's weight gradient:
// v = d sigma, X = the last trunk activation), and sum_p v[p] (its bias gradient).  HBM-bound: X is
// read once.  Thread `tid` owns the 16-byte pieces tid + 256 i (i < 8) of every tile it visits, i.e.
// always the same 4 features of the same in-tile sample, so it accumulates them privately; the 32
// samples of a tile are then folded with shuffles and the workgroups' partials are summed in a fixed
// order by wgrad_reduce_kernel.
__global__ __launch_bounds__(kThreads) void vecmat_kernel(const float* __restrict__ X, const float* __restrict__ vec,
                                                          int vec_stride, long P, long n_tiles,
                                                          float* __restrict__ part /* [G][257] */) {
    const int tid = threadIdx.x, lane = lane_id();
    const int m = lane & 31;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float vs = 0.f;
    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long p = tile * 32 + m;
        const float v = p < P ? vec[p * vec_stride] : 0.f;
        if (tid < 32) vs += v;
        const f32x4* blk = reinterpret_cast<const f32x4*>(X + tile * (32L * 256));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 x = blk[tid + kThreads * i];
            acc[i][0] = fmaf(v, x[0], acc[i][0]); acc[i][1] = fmaf(v, x[1], acc[i][1]);
            acc[i][2] = fmaf(v, x[2], acc[i][2]); acc[i][3] = fmaf(v, x[3], acc[i][3]);
        }
    }
    // fold the 32 samples (lanes with equal lane >> 5); piece tid + 256 i = (t*4+q)*64 + lane
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = acc[i][j];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) x += shfl_xor(x, o);
            acc[i][j] = x;
        }
    float vt = vs;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) vt += shfl_xor(vt, o);
    float* out = part + (long)blockIdx.x * 257;
    if (m == 0) {
        const int h = lane >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tq = ((tid + kThreads * i) >> 6);        // t*4 + q
            const int c = (tq >> 2) * 32 + (tq & 3) * 8 + 4 * h;
#pragma unroll
            for (int j = 0; j < 4; ++j) out[c + j] = acc[i][j];
        }
        if (tid == 0) out[256] = vt;
    }
}

// dW[c][k] = sum_p V[p][c] * X[p][k], db[c] = sum_p V[p][c] for c < 4: V row-major [P][4], X tile-native of width 128
// (rgb_linear's weight gradient: V = d_raw, whose fourth column -- d sigma -- is summed along and not reduced; X = the
// hidden layer of the views branch).  As vecmat_kernel: HBM-bound, X is read once, a thread owns the same 4 features of
// the same in-tile sample in every tile it visits; two tiles per iteration keep eight 16-byte loads in flight per thread.
// Partials [G][4][128] and [G][4] in ReduceJob's layout.
__global__ __launch_bounds__(kThreads) void rows4_kernel(const float* __restrict__ X, const float* __restrict__ V, long P,
                                                         long n_tiles, float* __restrict__ part_w, float* __restrict__ part_b) {
    const int tid = threadIdx.x, lane = lane_id();
    const int m = lane & 31;
    f32x4 acc[4][4];       // [piece][row]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 vs = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long tile = blockIdx.x; tile < n_tiles; tile += 2L * gridDim.x) {
        const long tile2 = tile + gridDim.x;
        const bool two = tile2 < n_tiles;
        const long p0 = tile * 32 + m, p1 = tile2 * 32 + m;
        const f32x4 v0 = p0 < P ? *reinterpret_cast<const f32x4*>(V + p0 * 4) : zero;
        const f32x4 v1 = (two && p1 < P) ? *reinterpret_cast<const f32x4*>(V + p1 * 4) : zero;
        if (tid < 32) vs += v0 + v1;
        const f32x4* blk0 = reinterpret_cast<const f32x4*>(X + tile * (32L * 128));
        const f32x4* blk1 = reinterpret_cast<const f32x4*>(X + (two ? tile2 : tile) * (32L * 128));
        f32x4 x0[4], x1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { x0[i] = blk0[tid + kThreads * i]; x1[i] = blk1[tid + kThreads * i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][c][j] = fmaf(v1[c], x1[i][j], fmaf(v0[c], x0[i][j], acc[i][c][j]));
    }
    // fold the 32 samples (lanes with equal lane >> 5); piece tid + 256 i = (t * 4 + q) * 64 + lane
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = acc[i][c][j];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) x += shfl_xor(x, o);
                acc[i][c][j] = x;
            }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float x = vs[c];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) x += shfl_xor(x, o);
        vs[c] = x;
    }
    float* ow = part_w + (long)blockIdx.x * (4 * 128);
    if (m == 0) {
        const int h = lane >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tq = (tid + kThreads * i) >> 6;          // t * 4 + q
            const int col = (tq >> 2) * 32 + (tq & 3) * 8 + 4 * h;
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(ow + c * 128 + col) = acc[i][c];
        }
        if (tid == 0) *reinterpret_cast<f32x4*>(part_b + (long)blockIdx.x * 4) = vs;
    }
}

// fixed-order sum over the G partials with four independent accumulators (four loads in flight per
// thread: the single-accumulator loop was latency-bound at ~1 TB/s)
__device__ __forceinline__ float sum_partials(const float* __restrict__ p, long stride, int G) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = 0;
    for (; g + 3 < G; g += 4) {
        const float a = p[(long)g * stride], b = p[(long)(g + 1) * stride];
        const float c = p[(long)(g + 2) * stride], d = p[(long)(g + 3) * stride];
        s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; g < G; ++g) s0 += p[(long)g * stride];
    return (s0 + s1) + (s2 + s3);
}

// out[n * ldo + col0 + k] = sum_g part[g][n][k]   (n < n_out, k < k_out), fixed order over g
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(
    const float* __restrict__ part_w, const float* __restrict__ part_b, int G, int BN, int BK, int n_out,
    int k_out, float* __restrict__ dW, int ldo, int col0, float* __restrict__ db) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nw = (long)n_out * k_out;
    if (idx < nw) {
        const int n = (int)(idx / k_out), k = (int)(idx % k_out);
        dW[(long)n * ldo + col0 + k] = sum_partials(part_w + (long)n * BK + k, (long)BN * BK, G);
        return;
    }
    const long j = idx - nw;
    if (db && j < n_out) db[j] = sum_partials(part_b + j, BN, G);
}

// dv[k] (+)= sum_g part[g][k] (k < 256), *dvsum (+)= sum_g part[g][256]
__global__ __launch_bounds__(256) void vecmat_reduce_kernel(const float* __restrict__ part, int G,
                                                            float* __restrict__ dv, float* __restrict__ dvsum,
                                                            int accumulate) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < 256) {
        const float v = sum_partials(part + k, 257, G);
        dv[k] = accumulate ? dv[k] + v : v;
    } else if (k == 256 && dvsum) {
        const float v = sum_partials(part + 256, 257, G);
        *dvsum = accumulate ? *dvsum + v : v;
    }
}

// ---- all reductions of one network pass in ONE launch -----------------------------------------------
// (13 dependent ~20 us launches between the GEMMs kept them from overlapping at their tails)
struct ReduceJob {
    const float* part_w; const float* part_b;
    float* dW; float* db;
    int G, BN, BK, n_out, k_out, ldo, col0;
    int block0;                 // first block of this job in the merged grid
};
constexpr int kMaxJobs = 16;
struct ReduceJobs { ReduceJob j[kMaxJobs]; int n; int accumulate; };

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(ReduceJobs jobs) {
    int q = 0;
#pragma unroll 1
    for (int i = 1; i < jobs.n; ++i)
        if ((int)blockIdx.x >= jobs.j[i].block0) q = i;
    const ReduceJob& J = jobs.j[q];
    const long idx = (long)(blockIdx.x - J.block0) * blockDim.x + threadIdx.x;
    const long nw = (long)J.n_out * J.k_out;
    if (idx < nw) {
        const int n = (int)(idx / J.k_out), k = (int)(idx % J.k_out);
        float* o = J.dW + (long)n * J.ldo + J.col0 + k;
        const float v = sum_partials(J.part_w + (long)n * J.BK + k, (long)J.BN * J.BK, J.G);
        *o = jobs.accumulate ? *o + v : v;
        return;
    }
    const long j = idx - nw;
    if (J.db && j < J.n_out) {
        const float v = sum_partials(J.part_b + j, J.BN, J.G);
        J.db[j] = jobs.accumulate ? J.db[j] + v : v;
    }
}

template <int WN, int WK, bool FAST>
int launch_wgrad_impl(const WgradArgs& a, int G, hipStream_t stream) {
    constexpr int BN = 2 * WN * 32, BK = 2 * WK * 32;
    const size_t lds = (size_t)2 * kMS * (BN + 4 + BK + 4) * sizeof(float);
    if (lds > 64 * 1024) SCN_LDS_OPT_IN((wgrad_kernel<WN, WK, FAST>), lds);
    hipLaunchKernelGGL((wgrad_kernel<WN, WK, FAST>), dim3(G), dim3(kThreads), lds, stream, a);
    return scn_launch_status();
}

template <int WN, int WK>
int launch_wgrad(const WgradArgs& a, int G, hipStream_t stream) {
    constexpr int BN = 2 * WN * 32, BK = 2 * WK * 32;
    if (a.a_tiled && a.b_tiled && a.n_load == BN && a.k_load == BK) return launch_wgrad_impl<WN, WK, true>(a, G, stream);
    return launch_wgrad_impl<WN, WK, false>(a, G, stream);
}



// How the 256 x 256 GEMMs (and the narrow ones with a tile-native dZ) multiply: 1 (default) = on three fp16 products per
// product with one power-of-two scale per operand and workgroup chunk, wherever the resident kernels left the chunk maxima
// (wgrad256_half.h, wgrad_half_narrow.h; error vs fp64 equal to the exact-fp32 MFMA kernel's); 0 = v_mfma_f32_32x32x2_f32
// (wgrad256.h, wgrad_tiles.h: the numerical yardstick, and what runs when no maxima were left).  Process-wide;
// SCNERF_WGRAD_ARITHMETIC=fp32 | half presets it, scnerf_wgrad_arithmetic() changes it.
int& wgrad_arithmetic() {
    static int mode = [] {
        const char* e = getenv("SCNERF_WGRAD_ARITHMETIC");
        return (e && (e[0] == 'f' || e[0] == '0')) ? 0 : 1;
    }();
    return mode;
}

// the 256 x 256 tile-native GEMMs of a pass in one launch on the fp32 MFMA, grid (chunks, jobs)
int launch_wgrad256(const wg256::Args& a, int G, hipStream_t stream) {
    constexpr int F = wg256::kSpread;
    SCN_LDS_OPT_IN((wg256::wgrad256_kernel<F>), wg256::kLdsBytes);
    hipLaunchKernelGGL((wg256::wgrad256_kernel<F>), dim3(G, a.n_jobs), dim3(wg256::kThreads), wg256::kLdsBytes, stream, a);
    return scn_launch_status();
}

// the same GEMMs on three fp16 products (wgrad256_half.h): needs the chunk maxima of both operands, [job][chunk]
int launch_wgrad256_half(const wg256::Args& a, int G, const float* amax_a, const float* amax_b, hipStream_t stream) {
    wg256h::Args h;
    for (int j = 0; j < a.n_jobs; ++j) h.job[j] = a.job[j];
    h.n_jobs = a.n_jobs;
    h.Ppad = a.Ppad;
    h.chunk = a.chunk;
    h.amax_a = amax_a;
    h.amax_b = amax_b;
    SCN_LDS_OPT_IN((wg256h::wgrad256_half_kernel<0>), wg256h::kLdsBytes);
    hipLaunchKernelGGL((wg256h::wgrad256_half_kernel<0>), dim3(G, a.n_jobs), dim3(wg256h::kThreads), wg256h::kLdsBytes,
                       stream, h);
    return scn_launch_status();
}

// narrower shapes with a tile-native dZ (wgrad_tiles.h)
template <int NA, int NB, bool B_ROWMAJOR>
int launch_wgrad_tiles(const wgt::Args& a, int G, hipStream_t stream) {
    constexpr unsigned lds = wgt::lds_bytes<NA, NB, B_ROWMAJOR>();
    if (lds > 64 * 1024) SCN_LDS_OPT_IN((wgt::wgrad_tiles_kernel<NA, NB, B_ROWMAJOR>), lds);
    hipLaunchKernelGGL((wgt::wgrad_tiles_kernel<NA, NB, B_ROWMAJOR>), dim3(G), dim3(wgt::kThreads), lds, stream, a);
    return scn_launch_status();
}

// the same shapes on three fp16 products (wgrad_half_narrow.h)
template <int NA, int NB, bool B_ROWMAJOR>
int launch_wgrad_half_narrow(const wgnh::Args& a, int G, hipStream_t stream, int jobs = 1) {
    SCN_LDS_OPT_IN((wgnh::wgrad_half_narrow_kernel<NA, NB, B_ROWMAJOR>), wgnh::kLdsBytes);
    hipLaunchKernelGGL((wgnh::wgrad_half_narrow_kernel<NA, NB, B_ROWMAJOR>), dim3(G, jobs), dim3(wgnh::kThreads), wgnh::kLdsBytes, stream, a);
    return scn_launch_status();
}

// two narrow GEMMs of one shape on the same X, queued by wgrad_gemm and launched together (grid (G, 2))
struct NarrowPair {
    wgnh::Args first;
    int n, k_load, G;
};

// where a narrow GEMM finds its operands' maxima (nullptr rows: stay on the fp32 MFMA)
struct NarrowScales {
    wgnh::Bound a, b;
    int n_coarse;
    long coarse_chunk;
};

struct Shape { int BN, BK; };

bool pick_shape(int n_load, int k_load, Shape* s) {
    // (BN, BK) of the six instantiations below
    if (n_load > 256 || k_load > 256) return false;
    if (n_load > 128) { s->BN = 256; s->BK = k_load > 128 ? 256 : (k_load > 64 ? 128 : 64); return true; }
    if (n_load > 64) { s->BN = 128; s->BK = k_load > 64 ? 256 : 64; return true; }
    s->BN = 64; s->BK = 128;
    return k_load <= 128;
}

}  // namespace

extern "C" long long scnerf_wgrad_workspace_floats(int n_load, int k_load, int n_chunks) {
    Shape s;
    if (!pick_shape(n_load, k_load, &s) || n_chunks < 1) return -1;
    return (long long)n_chunks * ((long long)s.BN * s.BK + s.BN);
}

namespace {
// GEMM into partials at `workspace`; fills `job` (the reduction that finishes it) and returns the floats used
int wgrad_gemm(const float* dz, int lda, int n_load, int n_out, int dz_tiled, const float* x, int ldb, int k_load,
               int k_out, int x_tiled, long long n_samples, int n_chunks, float* workspace, float* dW, int ldo,
               int col0, float* db, hipStream_t st, ReduceJob* job, long long* used, wg256::Args* batch = nullptr,
               const NarrowScales* narrow = nullptr, NarrowPair* pair = nullptr) {
    SCN_RETURN_IF(!dz || !x || !workspace || !dW || n_samples < 0 || n_chunks < 1, SCN_EINVAL);
    SCN_RETURN_IF(lda % 4 || ldb % 4 || n_load % 4 || k_load % 4 || n_out > n_load || k_out > k_load, SCN_EINVAL);
    SCN_RETURN_IF(((uintptr_t)dz | (uintptr_t)x | (uintptr_t)workspace) & 15, SCN_EINVAL);
    Shape s;
    SCN_RETURN_IF(!pick_shape(n_load, k_load, &s), SCN_ENOSUP);
    // a tile-native operand must fill the block tile exactly (its HBM block is the LDS image)
    SCN_RETURN_IF((dz_tiled && (lda != s.BN || n_load != lda)) || (x_tiled && (ldb != s.BK || k_load != ldb)), SCN_EINVAL);
    WgradArgs a;
    a.A = dz; a.lda = lda; a.n_load = n_load; a.a_tiled = dz_tiled;
    a.B = x; a.ldb = ldb; a.k_load = k_load; a.b_tiled = x_tiled;
    a.P = (long)n_samples;
    a.Ppad = scn::mlp::padded_samples(a.P);
    long chunk = (a.Ppad + n_chunks - 1) / n_chunks;
    chunk = (chunk + kMS - 1) / kMS * kMS;
    if (chunk == 0) chunk = kMS;
    a.chunk = chunk;
    const int G = n_chunks;
    a.part_w = workspace;
    a.part_b = a.part_w + (long)G * s.BN * s.BK;
    int rc;
    const bool both_tiled_256 = s.BN == 256 && s.BK == 256 && dz_tiled && x_tiled && n_load == 256 && k_load == 256;
    if (both_tiled_256) {
        // multi-GEMM kernel: queued into `batch` when the caller launches several at once
        wg256::Args one;
        wg256::Args* q = batch ? batch : &one;
        if (!batch) one.n_jobs = 0;
        SCN_RETURN_IF(q->n_jobs >= wg256::kMaxJobs, SCN_EINVAL);
        q->Ppad = a.Ppad;
        q->chunk = a.chunk;
        q->job[q->n_jobs++] = wg256::Job{dz, x, a.part_w, db ? a.part_b : nullptr};
        rc = batch ? 0 : launch_wgrad256(one, G, st);
    } else if (dz_tiled && n_load == lda && k_load == ldb && n_out <= n_load &&
               ((n_load == 256 && !x_tiled && (k_load == 64 || k_load == 128)) || (n_load == 128 && x_tiled && k_load == 256))) {
        if (narrow && narrow->a.amax && narrow->b.amax) {
            wgnh::Args t{dz, x, a.part_w, db ? a.part_b : nullptr, a.P, a.Ppad, a.chunk, narrow->a, narrow->b,
                         narrow->n_coarse, narrow->coarse_chunk, nullptr, nullptr, wgnh::Bound{nullptr, nullptr, nullptr},
                         nullptr, nullptr, nullptr, wgnh::Bound{nullptr, nullptr, nullptr}};
            if (pair && n_load == 256 && pair->n == 0) {
                pair->first = t; pair->n = 1; pair->k_load = k_load; pair->G = G;        // launched with its partner
                rc = 0;
            } else if (pair && n_load == 256 && pair->n == 1 && pair->k_load == k_load && pair->G == G && pair->first.B == x) {
                wgnh::Args both = pair->first;
                both.A_y1 = t.A; both.part_w_y1 = t.part_w; both.part_b_y1 = t.part_b; both.a_y1 = t.a;
                pair->n = 2;
                rc = k_load == 64 ? launch_wgrad_half_narrow<256, 64, true>(both, G, st, 2)
                                  : launch_wgrad_half_narrow<256, 128, true>(both, G, st, 2);
            } else if (n_load == 128) rc = launch_wgrad_half_narrow<128, 256, false>(t, G, st);
            else if (k_load == 64) rc = launch_wgrad_half_narrow<256, 64, true>(t, G, st);
            else rc = launch_wgrad_half_narrow<256, 128, true>(t, G, st);
        } else {
            wgt::Args t{dz, x, a.part_w, db ? a.part_b : nullptr, a.P, a.Ppad, a.chunk};
            if (n_load == 128) rc = launch_wgrad_tiles<128, 256, false>(t, G, st);
            else if (k_load == 64) rc = launch_wgrad_tiles<256, 64, true>(t, G, st);
            else rc = launch_wgrad_tiles<256, 128, true>(t, G, st);
        }
    } else if (s.BN == 256 && s.BK == 256) rc = launch_wgrad<4, 4>(a, G, st);
    else if (s.BN == 256 && s.BK == 128) rc = launch_wgrad<4, 2>(a, G, st);
    else if (s.BN == 256 && s.BK == 64) rc = launch_wgrad<4, 1>(a, G, st);
    else if (s.BN == 128 && s.BK == 256) rc = launch_wgrad<2, 4>(a, G, st);
    else if (s.BN == 128 && s.BK == 64) rc = launch_wgrad<2, 1>(a, G, st);
    else rc = launch_wgrad<1, 2>(a, G, st);
    SCN_RETURN_IF(rc != 0, rc);
    job->part_w = a.part_w; job->part_b = a.part_b; job->dW = dW; job->db = db;
    job->G = G; job->BN = s.BN; job->BK = s.BK; job->n_out = n_out; job->k_out = k_out; job->ldo = ldo; job->col0 = col0;
    job->block0 = 0;
    *used = (long long)G * ((long long)s.BN * s.BK + s.BN);
    return 0;
}
}  // namespace

extern "C" int scnerf_wgrad(const float* dz, int lda, int n_load, int n_out, int dz_tiled, const float* x,
                            int ldb, int k_load, int k_out, int x_tiled, long long n_samples, int n_chunks,
                            float* workspace, float* dW, int ldo, int col0, float* db, void* stream) {
    ReduceJob j;
    long long used;
    hipStream_t st = (hipStream_t)stream;
    const int rc = wgrad_gemm(dz, lda, n_load, n_out, dz_tiled, x, ldb, k_load, k_out, x_tiled, n_samples, n_chunks,
                              workspace, dW, ldo, col0, db, st, &j, &used);
    SCN_RETURN_IF(rc != 0, rc);
    const long total = (long)n_out * k_out + (db ? n_out : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(scn_ceil_div(total, 256)), dim3(256), 0, st, j.part_w, j.part_b,
                       j.G, j.BN, j.BK, n_out, k_out, dW, ldo, col0, db);
    return scn_launch_status();
}

namespace {
int vecmat_impl(const float* x_tiled256, const float* vec, int vec_stride, long long n_samples, int n_chunks,
                float* workspace, float* dv, float* dvsum, int accumulate, void* stream) {
    SCN_RETURN_IF(!x_tiled256 || !vec || !workspace || !dv || n_samples < 0 || n_chunks < 1 || vec_stride < 1, SCN_EINVAL);
    const long n_tiles = scn::mlp::padded_samples((long)n_samples) / 32;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(vecmat_kernel, dim3(n_chunks), dim3(kThreads), 0, st, x_tiled256, vec, vec_stride,
                       (long)n_samples, n_tiles, workspace);
    hipLaunchKernelGGL(vecmat_reduce_kernel, dim3(2), dim3(256), 0, st, workspace, n_chunks, dv, dvsum, accumulate);
    return scn_launch_status();
}
}  // namespace

extern "C" int scnerf_vecmat(const float* x_tiled256, const float* vec, int vec_stride, long long n_samples,
                             int n_chunks, float* workspace, float* dv, float* dvsum, void* stream) {
    return vecmat_impl(x_tiled256, vec, vec_stride, n_samples, n_chunks, workspace, dv, dvsum, 0, stream);
}

// ---- all weight gradients of one network (D=8, W=256, skip 4, view-dependent head; 3-D or 4-D point) ----
extern "C" int scnerf_nerf_param_count(int pt_dims) {
    return pt_dims == 4 ? scn::mlp::Var<4>::kNParams : scn::mlp::Var<3>::kNParams;
}

namespace {
// the GEMM slabs behind the vecmat partials stay 16-byte aligned (the persistent kernel stores them as float4)
long long vecmat_ws_floats(long long n_chunks) { return (257 * n_chunks + 3) / 4 * 4; }

// amax_x / amax_z (or nullptr): [8][n_chunks] chunk maxima of the X / dZ operands of the eight 256 x 256 GEMMs in the
// order they are queued below (layers 1 .. 7, feature_linear), left by the resident kernels: with them the eight run
// on three fp16 products (wgrad256_half.h) unless the arithmetic is switched to fp32.
// ev_before / ev_after (or nullptr): events recorded around that one launch (bench.py's per-kernel timing) -- arguments
// of THIS call, so that no exit path can leave them armed for a later one.
// The eight 256 x 256 GEMMs of a pass are ONE launch of (chunks, 8) workgroups, one per CU at a time: with an eighth of
// the chunks the narrow GEMMs use, 256 chunks become 32 x 8 = 256 workgroups -- one wave over the chip, every workgroup
// eight times as many samples -- and the partial slabs (256 KB each) shrink from 0.5 GB to 67 MB written and re-read.
int big_chunks(int n_chunks) { return (n_chunks >= 8 && n_chunks % 8 == 0) ? n_chunks / 8 : n_chunks; }

template <int PD>
int nerf_wgrad(const float* save, const float* grads, const float* d_raw, long long P, int n_chunks,
               float* workspace, float* g, int accumulate, void* stream, const float* amax_x = nullptr,
               const float* amax_z = nullptr, const float* scales = nullptr, hipEvent_t ev_before = nullptr,
               hipEvent_t ev_after = nullptr) {
    using namespace scn::mlp;
    using V = Var<PD>;
    const long long Ppad = scn::mlp::padded_samples(P);
    auto S = [&](int sec) { return save + (long long)sec * Ppad; };
    auto G = [&](int sec) { return grads + (long long)sec * Ppad; };
    auto act = [&](int l) { return S(kSaveAct + 256 * l); };
    auto dz = [&](int l) { return G(kGradDz + 256 * l); };
    constexpr int EW = V::kEW, IN = V::kInCh, SK = V::kSkipLd;
    int rc;
    ReduceJobs jobs;
    jobs.n = 0;
    jobs.accumulate = accumulate;
    wg256::Args big;                 // the eight 256 x 256 GEMMs go out as one launch
    big.n_jobs = 0;
    float* ws = workspace + vecmat_ws_floats(n_chunks);   // [0, 257 G) rounded up: the vecmat partials
    const int nb = big_chunks(n_chunks);                  // chunks of the eight 256 x 256 GEMMs
    hipStream_t st = (hipStream_t)stream;
    // the narrow GEMMs on three fp16 products: rows 8 .. 10 of the dZ maxima are dZ of the views layer, dZ of layer 0
    // and max(1, |point|) >= the encoded point (mlp_bwd_h3_kernel.h); the feature is bounded through its layer
    const bool half_narrow = amax_x && amax_z && scales && wgrad_arithmetic() == 1;
    const long coarse_chunk = (long)scnerf_wgrad_chunk_samples(P, nb);
    auto zrow = [&](int r) { return wgnh::Bound{amax_z + (long)r * nb, nullptr, nullptr}; };
    const NarrowScales ns_l0{zrow(9), zrow(10), nb, coarse_chunk};
    const NarrowScales ns_l5{zrow(4), zrow(10), nb, coarse_chunk};
    const NarrowScales ns_views{zrow(8),
                                wgnh::Bound{amax_x ? amax_x + 7L * nb : nullptr,
                                            scales ? scales + scn::h3::kLayerFeat * scn::h3::kScaleStride + scn::h3::kBoundA : nullptr,
                                            scales ? scales + scn::h3::kLayerFeat * scn::h3::kScaleStride + scn::h3::kBoundB : nullptr},
                                nb, coarse_chunk};
    const NarrowScales* narrow = nullptr;
    NarrowPair pair;                 // layer 0 and the skip columns of layer 5: same shape, same X -- one launch
    pair.n = 0;
#define SCN_WG(...)                                                                            \
    {                                                                                          \
        long long used__ = 0;                                                                  \
        rc = wgrad_gemm(__VA_ARGS__, st, &jobs.j[jobs.n], &used__, &big, narrow, &pair);       \
        if (rc != 0) return rc;                                                                \
        ws += used__;                                                                          \
        ++jobs.n;                                                                              \
    }
    // (dz, lda, n_load, n_out, tiled,  x, ldb, k_load, k_out, tiled,  P, chunks, ws, dW, ldo, col0, db)
    // layer 0: X = encoded points (row-major, IN valid of EW columns)
    narrow = half_narrow ? &ns_l0 : nullptr;
    SCN_WG(dz(0), 256, 256, 256, 1, S(kSaveEpts), EW, EW, IN, 0, P, n_chunks, ws, g + V::kW0, IN, 0, g + V::kB0)
    narrow = nullptr;
    for (int l = 1; l <= 7; ++l) {
        if (l == 5) {
            narrow = half_narrow ? &ns_l5 : nullptr;
            SCN_WG(dz(5), 256, 256, 256, 1, S(kSaveEpts), EW, EW, IN, 0, P, n_chunks, ws, g + V::trunk_w(5), SK, 0, nullptr)
            narrow = nullptr;
            SCN_WG(dz(5), 256, 256, 256, 1, act(4), 256, 256, 256, 1, P, nb, ws, g + V::trunk_w(5), SK, IN, g + V::trunk_b(5))
        } else {
            SCN_WG(dz(l), 256, 256, 256, 1, act(l - 1), 256, 256, 256, 1, P, nb, ws, g + V::trunk_w(l), 256, 0, g + V::trunk_b(l))
        }
    }
    if (pair.n == 1) {               // (a lone narrow GEMM of the pair's shape: never with this network, launched for safety)
        rc = pair.k_load == 64 ? launch_wgrad_half_narrow<256, 64, true>(pair.first, pair.G, st)
                               : launch_wgrad_half_narrow<256, 128, true>(pair.first, pair.G, st);
        if (rc != 0) return rc;
        pair.n = 0;                  // (flushed: a later GEMM of the pair's shape starts a new pair instead of re-launching this one)
    }
    // feature_linear; alpha_linear (one output row) = d sigma^T . act7 with d sigma = d_raw[:, 3]
    SCN_WG(G(kGradDfeat), 256, 256, 256, 1, act(7), 256, 256, 256, 1, P, nb, ws, g + V::kWF, 256, 0, g + V::kBF)
    {
        // (its partials [G][257] are finished by the merged reduction below: the 256 sums and the bias sum as two jobs)
        hipLaunchKernelGGL(vecmat_kernel, dim3(n_chunks), dim3(kThreads), 0, st, act(7), d_raw + 3, 4, (long)P, (long)(Ppad / 32), workspace);
        rc = scn_launch_status();
        if (rc != 0) return rc;
        ReduceJob& Jw = jobs.j[jobs.n++];
        Jw.part_w = workspace; Jw.part_b = nullptr; Jw.dW = g + V::kWA; Jw.db = nullptr;
        Jw.G = n_chunks; Jw.BN = 1; Jw.BK = 257; Jw.n_out = 1; Jw.k_out = 256; Jw.ldo = 256; Jw.col0 = 0; Jw.block0 = 0;
        ReduceJob& Jb = jobs.j[jobs.n++];
        Jb.part_w = workspace + 256; Jb.part_b = nullptr; Jb.dW = g + V::kBA; Jb.db = nullptr;
        Jb.G = n_chunks; Jb.BN = 1; Jb.BK = 257; Jb.n_out = 1; Jb.k_out = 1; Jb.ldo = 1; Jb.col0 = 0; Jb.block0 = 0;
    }
    // views layer: [feature | encoded direction]
    if (half_narrow) {
        // one launch for both X operands of the views layer: dZ is read (and cut) once (wgrad_half_narrow.h, WB2 = 32)
        const int Gv = n_chunks;
        long chunk = (Ppad + Gv - 1) / Gv;
        chunk = (chunk + kMS - 1) / kMS * kMS;
        float* part_w = ws;
        float* part_b = part_w + (long)Gv * 128 * 256;
        float* part_w2 = part_b + (long)Gv * 128;
        wgnh::Args t{G(kGradDzv), S(kSaveFeat), part_w, part_b, (long)P, (long)Ppad, chunk, ns_views.a, ns_views.b, nb, coarse_chunk,
                     S(kSaveEviews), part_w2, zrow(11)};
        SCN_LDS_OPT_IN((wgnh::wgrad_half_narrow_kernel<128, 256, false, 32>), wgnh::kLdsBytes);
        hipLaunchKernelGGL((wgnh::wgrad_half_narrow_kernel<128, 256, false, 32>), dim3(Gv), dim3(wgnh::kThreads), wgnh::kLdsBytes, st, t);
        rc = scn_launch_status();
        if (rc != 0) return rc;
        ReduceJob& J1 = jobs.j[jobs.n++];
        J1.part_w = part_w; J1.part_b = part_b; J1.dW = g + V::kWV; J1.db = g + V::kBV;
        J1.G = Gv; J1.BN = 128; J1.BK = 256; J1.n_out = 128; J1.k_out = 256; J1.ldo = 283; J1.col0 = 0; J1.block0 = 0;
        ReduceJob& J2 = jobs.j[jobs.n++];
        J2.part_w = part_w2; J2.part_b = nullptr; J2.dW = g + V::kWV; J2.db = nullptr;
        J2.G = Gv; J2.BN = 128; J2.BK = 32; J2.n_out = 128; J2.k_out = 27; J2.ldo = 283; J2.col0 = 256; J2.block0 = 0;
        ws += (long)Gv * (128 * 256 + 128 + 128 * 32);
    } else {
        SCN_WG(G(kGradDzv), 128, 128, 128, 1, S(kSaveFeat), 256, 256, 256, 1, P, n_chunks, ws, g + V::kWV, 283, 0, g + V::kBV)
        SCN_WG(G(kGradDzv), 128, 128, 128, 1, S(kSaveEviews), 32, 32, 27, 0, P, n_chunks, ws, g + V::kWV, 283, 256, nullptr)
    }
    // rgb_linear: dZ = d_raw[:, 0:3] (row-major), X = hidden of the views layer
    {
        // (rows4_kernel: the hidden layer is read once at the HBM rate; the general kernel pads three rows to a 64-row tile)
        const int Gr = n_chunks;
        float* part_w = ws;
        float* part_b = ws + (long)Gr * 4 * 128;
        hipLaunchKernelGGL(rows4_kernel, dim3(Gr), dim3(kThreads), 0, st, S(kSaveHv), d_raw, (long)P, (long)(Ppad / 32), part_w, part_b);
        rc = scn_launch_status();
        if (rc != 0) return rc;
        ReduceJob& J = jobs.j[jobs.n++];
        J.part_w = part_w; J.part_b = part_b; J.dW = g + V::kWRGB; J.db = g + V::kBRGB;
        J.G = Gr; J.BN = 4; J.BK = 128; J.n_out = 3; J.k_out = 128; J.ldo = 128; J.col0 = 0; J.block0 = 0;
        ws += (long)Gr * (4 * 128 + 4);
    }
#undef SCN_WG
    if (big.n_jobs > 0) {
        if (ev_before) SCN_HIP(hipEventRecord(ev_before, st));
        rc = (amax_x && amax_z && wgrad_arithmetic() == 1) ? launch_wgrad256_half(big, nb, amax_z, amax_x, st)
                                                            : launch_wgrad256(big, nb, st);
        if (rc != 0) return rc;
        if (ev_after) SCN_HIP(hipEventRecord(ev_after, st));
    }
    // one launch finishes all twelve GEMMs (fixed-order sums: deterministic)
    int blocks = 0;
    for (int i = 0; i < jobs.n; ++i) {
        jobs.j[i].block0 = blocks;
        const long total = (long)jobs.j[i].n_out * jobs.j[i].k_out + (jobs.j[i].db ? jobs.j[i].n_out : 0);
        blocks += (int)scn_ceil_div(total, 256);
    }
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), 0, st, jobs);
    return scn_launch_status();
}
}  // namespace

extern "C" int scnerf_nerf_wgrad(int pt_dims, const float* save, const float* grads, const float* d_raw,
                                 long long n_samples, int n_chunks, float* workspace, float* flat_grad,
                                 int accumulate, void* stream) {
    SCN_RETURN_IF(!save || !grads || !d_raw || !workspace || !flat_grad || n_samples < 1 || n_chunks < 1, SCN_EINVAL);
    SCN_RETURN_IF(pt_dims != 3 && pt_dims != 4, SCN_EINVAL);
    if (pt_dims == 3) return nerf_wgrad<3>(save, grads, d_raw, n_samples, n_chunks, workspace, flat_grad, accumulate, stream);
    return nerf_wgrad<4>(save, grads, d_raw, n_samples, n_chunks, workspace, flat_grad, accumulate, stream);
}

extern "C" int scnerf_wgrad_arithmetic(int mode) {
    if (mode == 0 || mode == 1) wgrad_arithmetic() = mode;
    return wgrad_arithmetic();
}

extern "C" int scnerf_wgrad256_chunks(int n_chunks) { return n_chunks < 1 ? -1 : big_chunks(n_chunks); }

extern "C" long long scnerf_wgrad_chunk_samples(long long n_samples, int n_chunks) {
    if (n_samples < 0 || n_chunks < 1) return -1;
    const long long Ppad = scn::mlp::padded_samples(n_samples);
    long long chunk = (Ppad + n_chunks - 1) / n_chunks;
    chunk = (chunk + kMS - 1) / kMS * kMS;
    return chunk == 0 ? kMS : chunk;
}

extern "C" int scnerf_nerf_wgrad_h3(int pt_dims, const float* save, const float* grads, const float* d_raw,
                                    long long n_samples, int n_chunks, float* workspace, float* flat_grad,
                                    int accumulate, const float* amax_x, const float* amax_z, const float* scales,
                                    void* ev_before, void* ev_after, void* stream) {
    SCN_RETURN_IF(!save || !grads || !d_raw || !workspace || !flat_grad || n_samples < 1 || n_chunks < 1, SCN_EINVAL);
    SCN_RETURN_IF(pt_dims != 3 && pt_dims != 4, SCN_EINVAL);
    if (pt_dims == 3) return nerf_wgrad<3>(save, grads, d_raw, n_samples, n_chunks, workspace, flat_grad, accumulate, stream, amax_x, amax_z, scales, (hipEvent_t)ev_before, (hipEvent_t)ev_after);
    return nerf_wgrad<4>(save, grads, d_raw, n_samples, n_chunks, workspace, flat_grad, accumulate, stream, amax_x, amax_z, scales, (hipEvent_t)ev_before, (hipEvent_t)ev_after);
}

// one narrow GEMM (256 x 64 / 256 x 128 with a row-major X, 128 x 256 with a tile-native X) on three fp16 products with
// the maxima given per coarse chunk of coarse_chunk samples (tests: accuracy against fp64); workspace as scnerf_wgrad
extern "C" int scnerf_wgrad_half_narrow(const float* dz_tiled, int n_load, const float* x, int k_load, int k_out,
                                        int x_tiled, long long n_samples, int n_chunks, float* workspace, float* dW,
                                        float* db, const float* amax_dz, const float* amax_x, int n_coarse,
                                        long long coarse_chunk, void* stream) {
    SCN_RETURN_IF(!amax_dz || !amax_x || n_coarse < 1 || coarse_chunk < 32, SCN_EINVAL);
    SCN_RETURN_IF(!((n_load == 256 && !x_tiled && (k_load == 64 || k_load == 128)) || (n_load == 128 && x_tiled && k_load == 256)), SCN_ENOSUP);
    ReduceJob j;
    long long used;
    hipStream_t st = (hipStream_t)stream;
    const NarrowScales ns{wgnh::Bound{amax_dz, nullptr, nullptr}, wgnh::Bound{amax_x, nullptr, nullptr}, n_coarse, (long)coarse_chunk};
    const int rc = wgrad_gemm(dz_tiled, n_load, n_load, n_load, 1, x, k_load, k_load, k_out, x_tiled, n_samples, n_chunks,
                              workspace, dW, k_out, 0, db, st, &j, &used, nullptr, &ns);
    SCN_RETURN_IF(rc != 0, rc);
    const long total = (long)n_load * k_out + (db ? n_load : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(scn_ceil_div(total, 256)), dim3(256), 0, st, j.part_w, j.part_b,
                       j.G, j.BN, j.BK, n_load, k_out, dW, k_out, 0, db);
    return scn_launch_status();
}

// one 256 x 256 GEMM on three fp16 products with the chunk maxima given (tests: accuracy against fp64)
extern "C" int scnerf_wgrad256_half(const float* dz_tiled, const float* x_tiled, long long n_samples, int n_chunks,
                                    float* workspace, float* dW, float* db, const float* amax_dz, const float* amax_x,
                                    void* stream) {
    SCN_RETURN_IF(!dz_tiled || !x_tiled || !workspace || !dW || !amax_dz || !amax_x || n_samples < 1 || n_chunks < 1, SCN_EINVAL);
    hipStream_t st = (hipStream_t)stream;
    wg256::Args one;
    one.n_jobs = 1;
    one.Ppad = scn::mlp::padded_samples(n_samples);
    one.chunk = scnerf_wgrad_chunk_samples(n_samples, n_chunks);
    float* part_w = workspace;
    float* part_b = workspace + (long)n_chunks * 256 * 256;
    one.job[0] = wg256::Job{dz_tiled, x_tiled, part_w, db ? part_b : nullptr};
    int rc = launch_wgrad256_half(one, n_chunks, amax_dz, amax_x, st);
    SCN_RETURN_IF(rc != 0, rc);
    const long total = 256L * 256 + (db ? 256 : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(scn_ceil_div(total, 256)), dim3(256), 0, st, part_w, db ? part_b : nullptr,
                       n_chunks, 256, 256, 256, 256, dW, 256, 0, db);
    return scn_launch_status();
}

long long scnerf_nerf_wgrad_workspace_floats(int n_chunks) {
    // every GEMM keeps its partials until the single reduction launch: 9 x (256 x 256), 2 x (256 x 128 | 64),
    // (128 x 256), (128 x 64), (64 x 128) blocks + the vecmat partials
    const long long G = n_chunks;
    auto blk = [&](long long bn, long long bk) { return G * (bn * bk + bn); };
    return vecmat_ws_floats(G) + 9 * blk(256, 256) + 2 * blk(256, 128) + blk(128, 256) + blk(128, 64) + blk(64, 128);
}
