// optim.hip -- fused Adam step over a flat fp32 segment (parameters, gradient and both moments are
// contiguous buffers: scnerf_amd.NeRF keeps all its tensors as views of one buffer).
//
// Replaces the per-tensor loop of f_custom_adam / torch.optim.Adam (/root/reference
// NeRF/create_nerf.py:199-254: ~8 element-wise launches for each of the 52 parameter tensors per step)
// with one HBM-bound launch per segment: 16 B read + 12 B written per element.  Arithmetic follows
// the reference's op order (file built with -ffp-contract=off):
//   g  = grad (+ weight_decay * p  for the decayed trailing tensors, :238-239)
//   m  = m * beta1 + g * (1 - beta1)
//   v  = v * beta2 + ((1 - beta2) * g) * g
//   p  = p + ((-lr / bc1) * m) / (sqrt(v) / sqrt(bc2) + eps)
#include <scn_wave.h>

#include "launch.h"
#include "scnerf_hip.h"

namespace {

using namespace scn;

// weight decay applies to elements [decay_lo, decay_hi) of the segment: the reference decays whole TENSORS (the last
// few of the stepped list), and a tensor boundary inside a flat segment need not be 16-byte aligned -- splitting the
// segment there would put an unaligned base into the vector path
struct AdamConst { float beta1, beta2, one_m_beta1, one_m_beta2, neg_step_size, sqrt_bc2, eps, weight_decay; long decay_lo, decay_hi; };

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamConst& c, long i) {
    if (c.weight_decay != 0.f && i >= c.decay_lo && i < c.decay_hi) g = g + c.weight_decay * p;
    m = m * c.beta1 + g * c.one_m_beta1;
    v = v * c.beta2 + (c.one_m_beta2 * g) * g;
    const float denom = sqrtf(v) / c.sqrt_bc2 + c.eps;
    p = p + (c.neg_step_size * m) / denom;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n,
                                                   AdamConst c) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = pp[j], b = mm[j], d = vv[j];
            adam_elem(a, gg[j], b, d, c, 4 * i + j);
            pp[j] = a; mm[j] = b; vv[j] = d;
        }
        reinterpret_cast<f32x4*>(p)[i] = pp;
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        adam_elem(p[i], g[i], m[i], v[i], c, i);
}

}  // namespace

extern "C" int scnerf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                long long n, double lr, double beta1, double beta2, double eps,
                                double weight_decay, long long step, void* stream) {
    return scnerf_adam_step_range(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, 0, n, step, stream);
}

extern "C" int scnerf_adam_step_range(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                      long long n, double lr, double beta1, double beta2, double eps,
                                      double weight_decay, long long decay_lo, long long decay_hi, long long step,
                                      void* stream) {
    SCN_RETURN_IF(!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1, SCN_EINVAL);
    SCN_RETURN_IF(decay_lo < 0 || decay_hi > n || decay_lo > decay_hi, SCN_EINVAL);
    if (n == 0) return 0;
    // 16-byte vector path needs aligned bases (segments of the flat buffers are)
    SCN_RETURN_IF((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0, SCN_EINVAL);
    // scalar prefactors in double like the reference's Python floats (:233-234, :252), then to fp32
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamConst c;
    c.beta1 = (float)beta1; c.beta2 = (float)beta2;
    c.one_m_beta1 = (float)(1.0 - beta1); c.one_m_beta2 = (float)(1.0 - beta2);
    c.neg_step_size = (float)(-(lr / bc1));
    c.sqrt_bc2 = (float)sqrt(bc2);
    c.eps = (float)eps;
    c.weight_decay = decay_hi > decay_lo ? (float)weight_decay : 0.f;
    c.decay_lo = (long)decay_lo;
    c.decay_hi = (long)decay_hi;
    const unsigned blocks = (unsigned)std::min<long long>(2048, (n / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, (long)n, c);
    return scn_launch_status();
}
