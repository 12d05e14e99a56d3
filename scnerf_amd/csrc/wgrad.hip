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
#include <scn_wave.h>

#include "launch.h"
#include "mlp_common.h"
#include "scnerf_hip.h"

namespace {

using namespace scn;

constexpr int kThreads = 256;
constexpr int kMS = 16;   // samples per LDS stage

struct WgradArgs {
    const float* A; int lda; int n_load;      // dZ  [P][lda], columns < n_load are read
    const float* B; int ldb; int k_load;      // X   [P][ldb], columns < k_load are read
    const float* vec; int vec_stride;         // optional v[p] = vec[p * vec_stride]
    long P;
    long chunk;                               // samples per workgroup (multiple of kMS)
    float* part_w;                            // [G][BN][BK]
    float* part_b;                            // [G][BN]
    float* part_v;                            // [G][BK + 1]  (last = sum v)
};

template <int WN, int WK, bool HAS_VEC>
__global__ __launch_bounds__(kThreads, 1) void wgrad_kernel(WgradArgs a) {
    constexpr int BN = 2 * WN * 32, BK = 2 * WK * 32;
    constexpr int STAGE = kMS * (BN + BK) + kMS;           // floats per LDS stage (+ vec)
    constexpr int A_F4 = kMS * BN / 4 / kThreads, B_F4 = kMS * BK / 4 / kThreads;
    static_assert(kMS * BN % (4 * kThreads) == 0 && kMS * BK % (4 * kThreads) == 0, "stage shape");
    float* lds = dynamic_lds<float>();
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const int wn = wave >> 1, wk = wave & 1;
    const int li = lane & 31, mh = lane >> 5;
    const long p_begin = (long)blockIdx.x * a.chunk;
    const long p_end = min(a.P, p_begin + a.chunk);
    const int n_stage = p_begin < p_end ? (int)((p_end - p_begin + kMS - 1) / kMS) : 0;

    f32x16 acc[WN][WK];
#pragma unroll
    for (int i = 0; i < WN; ++i)
#pragma unroll
        for (int j = 0; j < WK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[WN], vsum[WK], vtot = 0.f;
#pragma unroll
    for (int i = 0; i < WN; ++i) bsum[i] = 0.f;
#pragma unroll
    for (int j = 0; j < WK; ++j) vsum[j] = 0.f;

    f32x4 sa[A_F4], sb[B_F4];
    float sv = 0.f;
    auto issue = [&](int st) {
        const long p0 = p_begin + (long)st * kMS;
#pragma unroll
        for (int q = 0; q < A_F4; ++q) {
            const int e = (q * kThreads + tid) * 4;      // element index inside [kMS][BN]
            const int row = e / BN, col = e % BN;
            const long p = p0 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p < p_end && col < a.n_load) v = *reinterpret_cast<const f32x4*>(a.A + p * a.lda + col);
            sa[q] = v;
        }
#pragma unroll
        for (int q = 0; q < B_F4; ++q) {
            const int e = (q * kThreads + tid) * 4;
            const int row = e / BK, col = e % BK;
            const long p = p0 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p < p_end && col < a.k_load) v = *reinterpret_cast<const f32x4*>(a.B + p * a.ldb + col);
            sb[q] = v;
        }
        if (HAS_VEC && tid < kMS) {
            const long p = p0 + tid;
            sv = p < p_end ? a.vec[p * a.vec_stride] : 0.f;
        }
    };
    auto commit = [&](int buf) {
        float* s = lds + buf * STAGE;
#pragma unroll
        for (int q = 0; q < A_F4; ++q) *reinterpret_cast<f32x4*>(s + (q * kThreads + tid) * 4) = sa[q];
#pragma unroll
        for (int q = 0; q < B_F4; ++q) *reinterpret_cast<f32x4*>(s + kMS * BN + (q * kThreads + tid) * 4) = sb[q];
        if (HAS_VEC && tid < kMS) s[kMS * (BN + BK) + tid] = sv;
    };

    if (n_stage > 0) {
        issue(0);
        commit(0);
    }
    block_sync();
    for (int st = 0; st < n_stage; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < n_stage;
        if (more) issue(st + 1);
        sched_fence();          // the global loads stay at the head of the stage
        const float* As = lds + buf * STAGE;
        const float* Bs = As + kMS * BN;
        const float* Vs = Bs + kMS * BK;
        // operands of step s+1 are read while the 16 (WN x WK) MFMAs of step s run
        float av[2][WN], bv[2][WK], vv[2] = {0.f, 0.f};
        auto read_step = [&](int s2, int slot) {
            const int row = 2 * s2 + mh;
#pragma unroll
            for (int i = 0; i < WN; ++i) av[slot][i] = As[row * BN + (wn * WN + i) * 32 + li];
#pragma unroll
            for (int j = 0; j < WK; ++j) bv[slot][j] = Bs[row * BK + (wk * WK + j) * 32 + li];
            if (HAS_VEC) vv[slot] = Vs[row];
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
            if (wk == 0) {
#pragma unroll
                for (int i = 0; i < WN; ++i) bsum[i] += av[cur][i];
            }
            if (HAS_VEC && wn == 0) {
#pragma unroll
                for (int j = 0; j < WK; ++j) vsum[j] = fmaf(vv[cur], bv[cur][j], vsum[j]);
                if (wk == 0 && li == 0) vtot += vv[cur];
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
    if (HAS_VEC && wn == 0) {
#pragma unroll
        for (int j = 0; j < WK; ++j) {
            const float tot = vsum[j] + shfl_xor(vsum[j], 32);
            if (mh == 0) a.part_v[(long)blockIdx.x * (BK + 1) + (wk * WK + j) * 32 + li] = tot;
        }
        if (wk == 0) {
            const float tot = vtot + shfl_xor(vtot, 32);
            if (lane == 0) a.part_v[(long)blockIdx.x * (BK + 1) + BK] = tot;
        }
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
    const float* __restrict__ part_w, const float* __restrict__ part_b, const float* __restrict__ part_v, int G,
    int BN, int BK, int n_out, int k_out, float* __restrict__ dW, int ldo, int col0, float* __restrict__ db,
    float* __restrict__ dv, float* __restrict__ dvsum) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nw = (long)n_out * k_out;
    if (idx < nw) {
        const int n = (int)(idx / k_out), k = (int)(idx % k_out);
        dW[(long)n * ldo + col0 + k] = sum_partials(part_w + (long)n * BK + k, (long)BN * BK, G);
        return;
    }
    long j = idx - nw;
    if (db) {
        if (j < n_out) {
            db[j] = sum_partials(part_b + j, BN, G);
            return;
        }
        j -= n_out;
    }
    if (dv) {
        if (j < k_out) {
            dv[j] = sum_partials(part_v + j, BK + 1, G);
            return;
        }
        j -= k_out;
        if (j == 0 && dvsum) *dvsum = sum_partials(part_v + BK, BK + 1, G);
    }
}

template <int WN, int WK>
int launch_wgrad(const WgradArgs& a, int G, hipStream_t stream) {
    constexpr int BN = 2 * WN * 32, BK = 2 * WK * 32;
    const size_t lds = (size_t)2 * (kMS * (BN + BK) + kMS) * sizeof(float);
    if (a.vec) hipLaunchKernelGGL((wgrad_kernel<WN, WK, true>), dim3(G), dim3(kThreads), lds, stream, a);
    else hipLaunchKernelGGL((wgrad_kernel<WN, WK, false>), dim3(G), dim3(kThreads), lds, stream, a);
    return scn_launch_status();
}

struct Shape { int BN, BK; };

bool pick_shape(int n_load, int k_load, Shape* s) {
    // (BN, BK) of the five instantiations below
    if (n_load > 256 || k_load > 256) return false;
    if (n_load > 128) { s->BN = 256; s->BK = k_load > 64 ? 256 : 64; return true; }
    if (n_load > 64) { s->BN = 128; s->BK = k_load > 64 ? 256 : 64; return true; }
    s->BN = 64; s->BK = 128;
    return k_load <= 128;
}

}  // namespace

extern "C" long long scnerf_wgrad_workspace_floats(int n_load, int k_load, int n_chunks) {
    Shape s;
    if (!pick_shape(n_load, k_load, &s) || n_chunks < 1) return -1;
    return (long long)n_chunks * ((long long)s.BN * s.BK + s.BN + s.BK + 1);
}

extern "C" int scnerf_wgrad(const float* dz, int lda, int n_load, int n_out, const float* x, int ldb,
                            int k_load, int k_out, const float* vec, int vec_stride, long long n_samples,
                            int n_chunks, float* workspace, float* dW, int ldo, int col0, float* db,
                            float* dv, float* dvsum, void* stream) {
    SCN_RETURN_IF(!dz || !x || !workspace || !dW || n_samples < 0 || n_chunks < 1, SCN_EINVAL);
    SCN_RETURN_IF(lda % 4 || ldb % 4 || n_load % 4 || k_load % 4 || n_out > n_load || k_out > k_load, SCN_EINVAL);
    SCN_RETURN_IF(((uintptr_t)dz | (uintptr_t)x) & 15, SCN_EINVAL);
    Shape s;
    SCN_RETURN_IF(!pick_shape(n_load, k_load, &s), SCN_ENOSUP);
    WgradArgs a;
    a.A = dz; a.lda = lda; a.n_load = n_load;
    a.B = x; a.ldb = ldb; a.k_load = k_load;
    a.vec = vec; a.vec_stride = vec_stride;
    a.P = (long)n_samples;
    long chunk = (a.P + n_chunks - 1) / n_chunks;
    chunk = (chunk + kMS - 1) / kMS * kMS;
    if (chunk == 0) chunk = kMS;
    a.chunk = chunk;
    const int G = n_chunks;
    a.part_w = workspace;
    a.part_b = a.part_w + (long)G * s.BN * s.BK;
    a.part_v = a.part_b + (long)G * s.BN;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (s.BN == 256 && s.BK == 256) rc = launch_wgrad<4, 4>(a, G, st);
    else if (s.BN == 256 && s.BK == 64) rc = launch_wgrad<4, 1>(a, G, st);
    else if (s.BN == 128 && s.BK == 256) rc = launch_wgrad<2, 4>(a, G, st);
    else if (s.BN == 128 && s.BK == 64) rc = launch_wgrad<2, 1>(a, G, st);
    else rc = launch_wgrad<1, 2>(a, G, st);
    SCN_RETURN_IF(rc != 0, rc);
    const long total = (long)n_out * k_out + (db ? n_out : 0) + (dv ? k_out + 1 : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(scn_ceil_div(total, 256)), dim3(256), 0, st, a.part_w,
                       a.part_b, a.part_v, G, s.BN, s.BK, n_out, k_out, dW, ldo, col0, db,
                       vec ? dv : nullptr, vec ? dvsum : nullptr);
    return scn_launch_status();
}

// ---- all weight gradients of one standard NeRF (D=8, W=256, skip 4, view-dependent head) ----
namespace {
// flat parameter offsets, reference registration order (mirrors mlp_layout.PARAM_OFFSETS)
constexpr int kW0 = 0, kB0 = kW0 + 256 * 63;
constexpr int kTrunk1 = kB0 + 256;                       // layers 1..7: weight then bias
constexpr int trunk_w(int l) { return l <= 5 ? kTrunk1 + (l - 1) * (256 * 256 + 256) : kTrunk1 + 4 * (256 * 256 + 256) + (256 * 319 + 256) + (l - 6) * (256 * 256 + 256); }
constexpr int trunk_b(int l) { return trunk_w(l) + (l == 5 ? 256 * 319 : 256 * 256); }
constexpr int kWV = trunk_b(7) + 256, kBV = kWV + 128 * 283;
constexpr int kWF = kBV + 128, kBF = kWF + 256 * 256;
constexpr int kWA = kBF + 256, kBA = kWA + 256;
constexpr int kWRGB = kBA + 1, kBRGB = kWRGB + 3 * 128;
constexpr int kNParams = kBRGB + 3;
static_assert(kNParams == 595844, "parameter count of the standard NeRF");
}  // namespace

extern "C" int scnerf_nerf_param_count(void) { return kNParams; }

extern "C" int scnerf_nerf_wgrad(const float* save, const float* grads, const float* d_raw,
                                 long long n_samples, int n_chunks, float* workspace, float* flat_grad,
                                 void* stream) {
    using namespace scn::mlp;
    SCN_RETURN_IF(!save || !grads || !d_raw || !workspace || !flat_grad || n_samples < 1 || n_chunks < 1, SCN_EINVAL);
    const long long P = n_samples;
    auto S = [&](int sec) { return save + (long long)sec * P; };
    auto G = [&](int sec) { return grads + (long long)sec * P; };
    auto act = [&](int l) { return S(kSaveAct + 256 * l); };
    auto dz = [&](int l) { return G(kGradDz + 256 * l); };
    float* g = flat_grad;
    int rc;
#define SCN_WG(...)                          \
    rc = scnerf_wgrad(__VA_ARGS__, stream);  \
    if (rc != 0) return rc;
    // layer 0: X = encoded points (63 valid of 64 columns)
    SCN_WG(dz(0), 256, 256, 256, S(kSaveEpts), 64, 64, 63, nullptr, 0, P, n_chunks, workspace, g + kW0, 63, 0, g + kB0, nullptr, nullptr)
    for (int l = 1; l <= 7; ++l) {
        if (l == 5) {
            SCN_WG(dz(5), 256, 256, 256, S(kSaveEpts), 64, 64, 63, nullptr, 0, P, n_chunks, workspace, g + trunk_w(5), 319, 0, nullptr, nullptr, nullptr)
            SCN_WG(dz(5), 256, 256, 256, act(4), 256, 256, 256, nullptr, 0, P, n_chunks, workspace, g + trunk_w(5), 319, 63, g + trunk_b(5), nullptr, nullptr)
        } else {
            SCN_WG(dz(l), 256, 256, 256, act(l - 1), 256, 256, 256, nullptr, 0, P, n_chunks, workspace, g + trunk_w(l), 256, 0, g + trunk_b(l), nullptr, nullptr)
        }
    }
    // feature_linear (+ alpha_linear as the rank-1 side product with v = d sigma = d_raw[:, 3])
    SCN_WG(G(kGradDfeat), 256, 256, 256, act(7), 256, 256, 256, d_raw + 3, 4, P, n_chunks, workspace, g + kWF, 256, 0, g + kBF, g + kWA, g + kBA)
    // views layer: [feature | encoded direction]
    SCN_WG(G(kGradDzv), 128, 128, 128, S(kSaveFeat), 256, 256, 256, nullptr, 0, P, n_chunks, workspace, g + kWV, 283, 0, g + kBV, nullptr, nullptr)
    SCN_WG(G(kGradDzv), 128, 128, 128, S(kSaveEviews), 32, 32, 27, nullptr, 0, P, n_chunks, workspace, g + kWV, 283, 256, nullptr, nullptr, nullptr)
    // rgb_linear: dZ = d_raw[:, 0:3]
    SCN_WG(d_raw, 4, 4, 3, S(kSaveHv), 128, 128, 128, nullptr, 0, P, n_chunks, workspace, g + kWRGB, 128, 0, g + kBRGB, nullptr, nullptr)
#undef SCN_WG
    return 0;
}

extern "C" long long scnerf_nerf_wgrad_workspace_floats(int n_chunks) {
    return scnerf_wgrad_workspace_floats(256, 256, n_chunks);
}
