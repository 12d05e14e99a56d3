// aten_sum.h -- ATen's CPU row sum of a contiguous float row (SumKernel.cpp: 8 vector lanes, four
// interleaved cascade accumulators per lane), restated so that the pdf normalisers of the samplers
// round exactly like torch.sum on the reference's CPU path (oracle/scnerf_oracle.py aten_rowsum_f32
// documents and pins the order).  Executed by ONE lane; rows are <= a few hundred floats.
#pragma once
#include <scn_wave.h>

namespace scn {

__device__ inline int ceil_log2_aten(long x) {
    if (x <= 2) return 1;
    int l = 0;
    long v = x - 1;
    while (v > 0) { v >>= 1; ++l; }
    return l;
}

// four interleaved cascade accumulators over `size` groups; elem(i, k) = k-th of group i
template <typename Elem>
__device__ inline void multi_row_sum4(float out[4], Elem elem, long size) {
    const int level_power = max(4, ceil_log2_aten(size) / 4);
    const long level_step = 1L << level_power;
    const long level_mask = level_step - 1;
    float acc[4][4];
    for (int j = 0; j < 4; ++j)
        for (int k = 0; k < 4; ++k) acc[j][k] = 0.f;
    long i = 0;
    for (; i + level_step <= size;) {
        for (long j = 0; j < level_step; ++j, ++i)
            for (int k = 0; k < 4; ++k) acc[0][k] += elem(i, k);
        for (int j = 1; j < 4; ++j) {
            for (int k = 0; k < 4; ++k) {
                acc[j][k] += acc[j - 1][k];
                acc[j - 1][k] = 0.f;
            }
            const long mask = level_mask << (j * level_power);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < size; ++i)
        for (int k = 0; k < 4; ++k) acc[0][k] += elem(i, k);
    for (int j = 1; j < 4; ++j)
        for (int k = 0; k < 4; ++k) acc[0][k] += acc[j][k];
    for (int k = 0; k < 4; ++k) out[k] = acc[0][k];
}

template <typename Load>
__device__ inline float row_sum_ilp4(Load load, long size) {
    const long size_ilp = size / 4;
    float p[4];
    multi_row_sum4(p, [&](long i, int k) { return load(i * 4 + k); }, size_ilp);
    for (long i = size_ilp * 4; i < size; ++i) p[0] += load(i);
    for (int k = 1; k < 4; ++k) p[0] += p[k];
    return p[0];
}

__device__ inline float aten_rowsum(const float* w, int m) {
    constexpr int V = 8;
    if (m < V) return row_sum_ilp4([&](long i) { return w[i]; }, m);
    const int nv = m / V;
    float lanes[V];
    for (int v = 0; v < V; ++v) lanes[v] = row_sum_ilp4([&](long i) { return w[i * V + v]; }, nv);
    float acc = 0.f;
    for (int k = nv * V; k < m; ++k) acc += w[k];
    for (int v = 0; v < V; ++v) acc += lanes[v];
    return acc;
}

}  // namespace scn
