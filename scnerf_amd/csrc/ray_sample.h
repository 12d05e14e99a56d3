// ray_sample.h -- the hierarchical (inverse-CDF) sampler's per-ray pieces, shared by the stand-alone kernels
// (sampling.hip: sample_pdf, fine_sample) and the fused fine stage (mlp_fwd_h3_kernel.h, STAGE 2), so that the two routes
// are the SAME instructions and give bit-identical depths and indices.
//
//   sample_pdf            /root/reference NeRF/render.py:417-460   (cdf :421-426, search :444, inverse cdf :446-458)
//   the fine stage's use  /root/reference NeRF/render.py:269-277   (mid points, weights[1:-1], detach, sort of the cat)
//
// One 64-lane wave owns one ray; the per-ray arrays live in LDS, private to the wave.  The pieces order their LDS traffic
// with block_sync(): EVERY wave of the workgroup must call them the same number of times.
#pragma once
#include <scn_wave.h>

#include "aten_sum.h"

namespace scn {
namespace ray {

// cdf[0..nb) from weights w_in[0..nb-1) (already offset); s_w is scratch of >= nb floats.
// Up to 64 knots (the render path's 63) the wave works side by side: ATen's row sum with its eight vector lanes on eight GPU
// lanes (aten_rowsum_wave), one division per lane, and the fp64 running sum -- sequential, as torch.cumsum's -- fed from the
// lanes' registers (read_lane) instead of one lane walking the row through LDS with a round trip per element.  Same
// operations in the same order on the same numbers: bit-identical to the one-lane form, which larger rows keep.
__device__ inline void build_cdf(const float* w_in, int nb, float* s_w, float* s_cdf, int lane) {
    const int m = nb - 1;
    for (int k = lane; k < m; k += kWave) s_w[k] = w_in[k] + 1e-5f;
    block_sync();
    if (nb <= kWave) {
        const float tot = aten_rowsum_wave(s_w, m, lane);
        const double pd = lane < m ? (double)(s_w[lane] / tot) : 0.0;
        double run = 0.0;
        float mine = 0.f;                                   // lane k ends up with cdf[k]; cdf[0] = 0
#pragma unroll 4
        for (int k = 0; k < m; ++k) {
            run += read_lane(pd, k);
            if (lane == k + 1) mine = (float)run;
        }
        if (lane < nb) s_cdf[lane] = mine;
    } else if (lane == 0) {
        const float tot = aten_rowsum(s_w, m);
        double run = 0.0;
        s_cdf[0] = 0.f;
#pragma unroll 1
        for (int k = 0; k < m; ++k) {
            const float pdf = s_w[k] / tot;
            run += (double)pdf;
            s_cdf[k + 1] = (float)run;
        }
    }
    block_sync();
}

// upper bound (count of cdf entries <= u) for the ns samples of this ray; inds into s_ind.  Up to 64 knots: a knot per
// lane, one v_cmp + ballot + s_bcnt1 per sample, the sample's u broadcast from the registers of the lane that holds it.
__device__ inline void search_right(const float* s_cdf, int nb, const float* s_u, int ns, int* s_ind,
                                    int lane, bool side_left) {
    if (nb <= kWave) {
        const float c = lane < nb ? s_cdf[lane] : 0.f;
#pragma unroll 1
        for (int j0 = 0; j0 < ns; j0 += kWave) {
            const int nj = min(kWave, ns - j0);
            const float ub = lane < nj ? s_u[j0 + lane] : 0.f;
            int mine = 0;
#pragma unroll 4
            for (int jj = 0; jj < nj; ++jj) {
                const float uq = read_lane(ub, jj);
                const bool le = side_left ? (c < uq) : (c <= uq);
                const int cnt = popcount64(ballot(lane < nb && le));
                if (lane == jj) mine = cnt;
            }
            if (lane < nj) s_ind[j0 + lane] = mine;
        }
    } else {
        for (int j = lane; j < ns; j += kWave) {
            const float uq = s_u[j];
            int lo = 0, hi = nb;  // first index with cdf[idx] > u  (>= for side_left)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const float c = s_cdf[mid];
                const bool go_right = side_left ? (c < uq) : (c <= uq);
                if (go_right) lo = mid + 1; else hi = mid;
            }
            s_ind[j] = lo;
        }
    }
}

__device__ __forceinline__ float invert_cdf(const float* s_cdf, const float* s_bins, int nb,
                                            float u, int ind) {
    const int below = max(0, ind - 1);
    const int above = min(nb - 1, ind);
    const float c0 = s_cdf[below], c1 = s_cdf[above];
    const float b0 = s_bins[below], b1 = s_bins[above];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.f;
    const float t = (u - c0) / denom;
    return b0 + t * (b1 - b0);
}

__device__ __forceinline__ bool total_less(float a, int ia, float b, int ib) {
    // order used by the merge: numbers ascending, NaN last (as torch.sort), ties by position
    const bool an = a != a, bn = b != b;
    if (an || bn) return (!an && bn) || (an && bn && ia < ib);
    return (a < b) || (a == b && ia < ib);
}

// ---- the merge: s_sorted = torch.sort of the tot values in s_all (numbers ascending, -0 and +0 equal, NaN last, ties by
// position: total_less).  Up to 256 values: a bitonic network over (key, position) pairs held four per lane -- 36 stages of
// compare-exchange, the strides 1 and 2 inside a lane, the larger ones through shfl_xor -- ~650 instructions where counting
// ranks pair by pair took 6 500 (ballot counts) or 4 600 plus 37 000 LDS round trips (every lane walking every value).
// The key orders like the float (sign-flipped bits, both zeros on one key, NaN above +inf); the position breaks ties and
// finds the value again, so the merged row carries the original bit patterns.
__device__ __forceinline__ unsigned long long merge_key(float v, int pos) {
    unsigned k;
    if (v != v) {
        k = 0xffffffffu;
    } else {
        const unsigned b = v == 0.f ? 0u : __float_as_uint(v);
        k = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    }
    return ((unsigned long long)k << 32) | (unsigned)pos;
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
    const unsigned lo = (unsigned)shfl_xor((int)(unsigned)v, mask), hi = (unsigned)shfl_xor((int)(unsigned)(v >> 32), mask);
    return ((unsigned long long)hi << 32) | lo;
}

template <int K, int J>
__device__ __forceinline__ void bitonic_stage(unsigned long long (&key)[4], int lane) {
    if constexpr (J >= 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned long long other = shfl_xor_u64(key[r], J >> 2);
            const int i = lane * 4 + r;
            const bool keep_min = ((i & J) == 0) == ((i & K) == 0);
            const bool other_less = other < key[r];
            key[r] = (keep_min == other_less) ? other : key[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if ((r & J) != 0) continue;
            const int i = lane * 4 + r;
            const bool asc = (i & K) == 0;
            const unsigned long long a = key[r], b = key[r | J];
            const bool swap = (b < a) == asc;
            key[r] = swap ? b : a;
            key[r | J] = swap ? a : b;
        }
    }
}
template <int K, int J>
__device__ __forceinline__ void bitonic_phase(unsigned long long (&key)[4], int lane) {
    bitonic_stage<K, J>(key, lane);
    if constexpr (J > 1) bitonic_phase<K, J / 2>(key, lane);
}
template <int K>
__device__ __forceinline__ void bitonic_sort256(unsigned long long (&key)[4], int lane) {
    bitonic_phase<K, K / 2>(key, lane);
    if constexpr (K < 256) bitonic_sort256<K * 2>(key, lane);
}

__device__ inline void rank_merge(const float* s_all, int tot, float* s_sorted, int lane) {
    if (tot > 4 * kWave) {
        for (int e = lane; e < tot; e += kWave) {
            const float v = s_all[e];
            int rank = 0;
#pragma unroll 4
            for (int j = 0; j < tot; ++j) rank += total_less(s_all[j], j, v, e) ? 1 : 0;
            s_sorted[rank] = v;
        }
        return;
    }
    unsigned long long key[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = lane * 4 + r;          // (slots behind the last value: the NaN key at a later position -- they sort last)
        key[r] = i < tot ? merge_key(s_all[i], i) : ((0xffffffffull << 32) | (unsigned)i);
    }
    bitonic_sort256<2>(key, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = lane * 4 + r;
        if (i < tot) s_sorted[i] = s_all[(int)(unsigned)key[r]];
    }
}

// LDS floats one ray of the fine sampler needs: w, cdf, bins (sc - 1 each), u, indices (sf each), the unsorted and the
// sorted depths (sc + sf each)
__host__ __device__ constexpr int fine_sample_lds_floats(int sc, int sf) { return 3 * (sc - 1) + 2 * sf + 2 * (sc + sf); }

// The fine stage's sampling of ONE ray by one wave (NeRF/render.py:269-277): bins = mid points of the coarse depths,
// weights[1:-1], inverse cdf at u, merged with the coarse depths by rank.  `lds`: fine_sample_lds_floats(sc, sf) floats
// of this wave.  Writes (when `live`) z_f [sc + sf], pts_f [sc + sf][3] = o + d z, z_samples [sf], z_std [1] and, if given,
// inds [sf], cdf_out [sc - 1] -- all pointers already at the ray's row.  -> the sorted depths in LDS (sc + sf floats; valid
// after the call, which ends with a block_sync()).
__device__ inline const float* fine_sample_ray(const float* __restrict__ ray_row, const float* __restrict__ zc,
                                               const float* __restrict__ wc, const float* __restrict__ u_row, int sc, int sf,
                                               float* lds, int lane, bool live, float* __restrict__ z_f,
                                               float* __restrict__ pts_f, float* __restrict__ z_samples,
                                               float* __restrict__ z_std, int64_t* __restrict__ inds,
                                               float* __restrict__ cdf_out) {
    const int nb = sc - 1, tot = sc + sf;
    float* s_w = lds;
    float* s_cdf = s_w + nb;
    float* s_bins = s_cdf + nb;
    float* s_u = s_bins + nb;
    int* s_ind = reinterpret_cast<int*>(s_u + sf);
    float* s_all = reinterpret_cast<float*>(s_ind + sf);  // [tot] unsorted: z_c then samples
    float* s_sorted = s_all + tot;                        // [tot]
    for (int k = lane; k < sc; k += kWave) s_all[k] = zc[k];
    for (int j = lane; j < sf; j += kWave) s_u[j] = u_row[j];
    block_sync();
    for (int k = lane; k < nb; k += kWave) s_bins[k] = 0.5f * (s_all[k + 1] + s_all[k]);
    // weights[..., 1:-1]  (NeRF/render.py:270)
    build_cdf(wc + 1, nb, s_w, s_cdf, lane);
    search_right(s_cdf, nb, s_u, sf, s_ind, lane, false);
    block_sync();
    double part = 0.0;
    for (int j = lane; j < sf; j += kWave) {
        const float zs = invert_cdf(s_cdf, s_bins, nb, s_u[j], s_ind[j]);
        s_all[sc + j] = zs;
        part += (double)zs;
        if (live) {
            z_samples[j] = zs;
            if (inds) inds[j] = (int64_t)s_ind[j];
        }
    }
    if (live && cdf_out)
        for (int k = lane; k < nb; k += kWave) cdf_out[k] = s_cdf[k];
    // population std of the new samples (two-pass, fp64)
    for (int o = 32; o > 0; o >>= 1) part += shfl_xor(part, o);
    const double mean = part / (double)sf;
    double var = 0.0;
    block_sync();
    for (int j = lane; j < sf; j += kWave) {
        const double dlt = (double)s_all[sc + j] - mean;
        var += dlt * dlt;
    }
    for (int o = 32; o > 0; o >>= 1) var += shfl_xor(var, o);
    if (live && lane == 0) *z_std = (float)sqrt(var / (double)sf);
    // rank merge of the sc + sf depths (values only matter; equals torch.sort of the cat)
    rank_merge(s_all, tot, s_sorted, lane);
    block_sync();
    if (live) {
        const float ox = ray_row[0], oy = ray_row[1], oz = ray_row[2], dx = ray_row[3], dy = ray_row[4], dz = ray_row[5];
        for (int e = lane; e < tot; e += kWave) {
            const float z = s_sorted[e];
            z_f[e] = z;
            pts_f[e * 3 + 0] = ox + dx * z;
            pts_f[e * 3 + 1] = oy + dy * z;
            pts_f[e * 3 + 2] = oz + dz * z;
        }
    }
    return s_sorted;
}

}  // namespace ray
}  // namespace scn
