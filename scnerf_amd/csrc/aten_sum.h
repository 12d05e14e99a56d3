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

// One "vector lane" of the ATen sum: elements w[i * stride], i < size.  Four interleaved partial sums
// (ILP 4), each a cascade of four levels: level 0 takes level_step groups, then folds upwards (level j is
// folded into j+1 whenever the group counter reaches a multiple of level_step^(j+1)); the tail groups go to
// level 0; levels are added 1, 2, 3 into 0; then the ungrouped tail, then p0 += p1 += p2 += p3.
// Written with named scalars and rolled loops: as private arrays with run-time level indices this cost the
// samplers 248 VGPRs plus scratch (one wave per SIMD).
__device__ inline float aten_lane_sum(const float* w, int stride, int size) {
    const int size_ilp = size >> 2;
    const int level_power = max(4, ceil_log2_aten(size_ilp) / 4);
    const int level_step = 1 << level_power;
    const int level_mask = level_step - 1;
    float a00 = 0.f, a01 = 0.f, a02 = 0.f, a03 = 0.f;      // level 0, ILP slot 0..3
    float a10 = 0.f, a11 = 0.f, a12 = 0.f, a13 = 0.f;
    float a20 = 0.f, a21 = 0.f, a22 = 0.f, a23 = 0.f;
    float a30 = 0.f, a31 = 0.f, a32 = 0.f, a33 = 0.f;
    int i = 0;
#pragma unroll 1
    while (i + level_step <= size_ilp) {
#pragma unroll 1
        for (int j = 0; j < level_step; ++j, ++i) {
            a00 += w[(i * 4 + 0) * stride];
            a01 += w[(i * 4 + 1) * stride];
            a02 += w[(i * 4 + 2) * stride];
            a03 += w[(i * 4 + 3) * stride];
        }
        a10 += a00; a11 += a01; a12 += a02; a13 += a03;
        a00 = a01 = a02 = a03 = 0.f;
        if ((i & (level_mask << level_power)) == 0) {
            a20 += a10; a21 += a11; a22 += a12; a23 += a13;
            a10 = a11 = a12 = a13 = 0.f;
            if ((i & (level_mask << (2 * level_power))) == 0) {
                a30 += a20; a31 += a21; a32 += a22; a33 += a23;
                a20 = a21 = a22 = a23 = 0.f;
            }
        }
    }
#pragma unroll 1
    for (; i < size_ilp; ++i) {
        a00 += w[(i * 4 + 0) * stride];
        a01 += w[(i * 4 + 1) * stride];
        a02 += w[(i * 4 + 2) * stride];
        a03 += w[(i * 4 + 3) * stride];
    }
    a00 += a10; a01 += a11; a02 += a12; a03 += a13;
    a00 += a20; a01 += a21; a02 += a22; a03 += a23;
    a00 += a30; a01 += a31; a02 += a32; a03 += a33;
#pragma unroll 1
    for (int k = size_ilp * 4; k < size; ++k) a00 += w[k * stride];
    a00 += a01;
    a00 += a02;
    a00 += a03;
    return a00;
}

// torch.sum over a contiguous float row of m elements (m < 8: one lane over the row; else 8 vector lanes
// over the strided elements, the row tail, then the lanes in order)
__device__ inline float aten_rowsum(const float* w, int m) {
    if (m < 8) return aten_lane_sum(w, 1, m);
    const int nv = m >> 3;
    float acc = 0.f;
#pragma unroll 1
    for (int k = nv * 8; k < m; ++k) acc += w[k];
#pragma unroll 1
    for (int v = 0; v < 8; ++v) acc += aten_lane_sum(w + v, 8, nv);
    return acc;
}

// aten_rowsum by a whole wave: the eight vector lanes' sums are formed by eight GPU lanes side by side -- each with ATen's
// own sequence of additions -- and then added in order by every lane alike: bit-identical to aten_rowsum at about an
// eighth of its latency (the serial version is one lane walking the row through LDS, a round trip per element).  Every
// lane of the wave must call it; the result is the same in every lane.
__device__ inline float aten_rowsum_wave(const float* w, int m, int lane) {
    if (m < 8) return aten_lane_sum(w, 1, m);
    const int nv = m >> 3;
    float acc = 0.f;
#pragma unroll 1
    for (int k = nv * 8; k < m; ++k) acc += w[k];
    const float mine = aten_lane_sum(w + (lane & 7), 8, nv);       // lanes 0 .. 7: the eight vector lanes (the others repeat them)
#pragma unroll
    for (int v = 0; v < 8; ++v) acc += read_lane(mine, v);
    return acc;
}

}  // namespace scn
