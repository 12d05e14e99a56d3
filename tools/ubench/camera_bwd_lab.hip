// camera_bwd_lab.hip -- LAB (never part of the product build): the product's camera_rays_bwd_kernel, statement for statement, with a
// per-ray DUMP of what it computed written at the very end (after the atomics and the wave reduction): the intrinsics it read, the
// rotation columns, the raw direction and its norm, the transformed direction gradient, the two dot products that feed the
// intrinsics' gradient, and the hardware id of the wave (XCC, SE, CU, SIMD).  tools/flaky_probe5.py calls it again and again on
// identical inputs while a second process keeps the GPU busy and compares dumps bit for bit: which value moves first, in which
// lanes, on which CU.  The product's source is included, not copied; only the kernel below repeats the product kernel's body.
// LAB_CAMERA_SOURCE: the product's camera_rays.hip, or a sed-made variant of it (tools/ubench/build_camera_bwd_lab.sh: the four noise-grid taps
// as three one-word loads each instead of one three-word load); LAB_WARM 1: every lane touches the direction grid once before anything else
#ifndef LAB_CAMERA_SOURCE
#define LAB_CAMERA_SOURCE "../../scnerf_amd/csrc/camera_rays.hip"
#endif
#ifndef LAB_WARM
#define LAB_WARM 0
#endif
#include LAB_CAMERA_SOURCE

namespace {

constexpr int kDump = 32;     // floats per ray

__global__ __launch_bounds__(256) void camera_rays_bwd_dump_kernel(CamArgs a, const float* __restrict__ g_o, const float* __restrict__ g_d,
                                                                   float* acc, float* d_grid_o, float* d_grid_d, int lds_slots,
                                                                   float* __restrict__ dump) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float* lacc = dynamic_lds<float>();
    if (lds_slots > 0) {
        for (int k = threadIdx.x; k < lds_slots * 12; k += blockDim.x) lacc[k] = 0.f;
        block_sync();
    }
    float gk[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float keep[kDump];
#pragma unroll
    for (int k = 0; k < kDump; ++k) keep[k] = 0.f;
    float warm = 0.f;
    if (LAB_WARM && a.grid_d) warm = a.grid_d[threadIdx.x & 3] * 0.f;        // not foldable: the grid may hold a NaN
    if (i < a.n) {
        const Intrinsics K = intrinsics_of(a);
        RayFwd f;
        Vec3 o, d;
        ray_forward(a, i, K, &f, &o, &d);
        const Vec3 go = g_o ? v3(g_o[(size_t)i * 3], g_o[(size_t)i * 3 + 1], g_o[(size_t)i * 3 + 2]) : v3(0, 0, 0);
        Vec3 gr = g_d ? v3(g_d[(size_t)i * 3], g_d[(size_t)i * 3 + 1], g_d[(size_t)i * 3 + 2]) : v3(0, 0, 0);
        keep[26] = gr.x; keep[27] = gr.y; keep[28] = gr.z;
        if (a.grid_d) {
            const float den = a.select ? f.nrm : f.nrm + 1e-10f;
            Vec3 g = (1.f / den) * gr;
            if (f.nrm > 0.f) g = g - (dot(gr, f.rd_raw) / (den * den * f.nrm)) * f.rd_raw;
            gr = g;
            if (d_grid_d) scatter_grid(d_grid_d, a.gw, f.taps, a.scale_d * gr);
        }
        if (a.grid_o && d_grid_o) scatter_grid(d_grid_o, a.gw, f.taps, a.scale_o * go);
        const int slot = a.extrinsic ? (a.n_ext == 1 ? 0 : i) : f.cam;
        const Vec3 dd = f.dirs;
        const float term[12] = {gr.x * dd.x, gr.y * dd.x, gr.z * dd.x, gr.x * dd.y, gr.y * dd.y, gr.z * dd.y,
                                gr.x * dd.z, gr.y * dd.z, gr.z * dd.z, go.x, go.y, go.z};
        if (lds_slots > 0) {
            float* ar = lacc + slot * 12;
#pragma unroll
            for (int k = 0; k < 12; ++k) atomic_add(ar + k, term[k]);
        } else {
            float* ar = acc + 4 + (size_t)slot * 12;
#pragma unroll
            for (int k = 0; k < 12; ++k) atomic_add(ar + k, term[k]);
        }
        const float gdx = dot(gr, f.R.x), gdy = dot(gr, f.R.y);
        gk[0] = gdx * (-(f.x - K.cx) / (K.fx * K.fx));
        gk[2] = gdx * (-1.f / K.fx);
        gk[1] = gdy * ((f.y - K.cy) / (K.fy * K.fy));
        gk[3] = gdy * (1.f / K.fy);
        keep[0] = K.fx; keep[1] = K.fy; keep[2] = K.cx; keep[3] = K.cy;
        keep[4] = f.x; keep[5] = f.y;
#ifdef LAB_TRACE
#pragma unroll
        for (int k = 0; k < 6; ++k) keep[k] = f.dbg[k];      // tap (y0, x0) as loaded (x, y, z), tap (y0, x1).y, wx0, the first blend's y
#endif
        keep[6] = dd.x; keep[7] = dd.y; keep[8] = dd.z;
        keep[9] = f.R.x.x; keep[10] = f.R.x.y; keep[11] = f.R.x.z;
        keep[12] = f.R.y.x; keep[13] = f.R.y.y; keep[14] = f.R.y.z;
        keep[15] = f.rd_raw.x; keep[16] = f.rd_raw.y; keep[17] = f.rd_raw.z;
        keep[18] = f.nrm;
        keep[19] = gr.x; keep[20] = gr.y; keep[21] = gr.z;
        keep[22] = gdx; keep[23] = gdy;
        keep[24] = gk[0]; keep[25] = gk[1];
        keep[29] = (float)f.cam + warm;
    }
    if (lds_slots > 0) {
        block_sync();
        for (int k = threadIdx.x; k < lds_slots * 12; k += blockDim.x) {
            const float v = lacc[k];
            if (v != 0.f) atomic_add(acc + 4 + k, v);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = gk[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += shfl_xor(v, o);
        if (lane_id() == 0) atomic_add(acc + k, v);
        if (k == 0) keep[30] = v;                          // the wave's sum of gk[0], as every lane holds it after the butterfly
    }
    // hardware id: HW_REG_HW_ID (4) and HW_REG_XCC_ID (20), 32 bits each
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    keep[31] = __uint_as_float(((xcc & 0xf) << 28) | (hw & 0x0fffffff));
    if (i < a.n) {
#pragma unroll
        for (int k = 0; k < kDump; ++k) dump[(size_t)i * kDump + k] = keep[k];
    }
}

}  // namespace

// same arguments as scnerf_camera_rays_bwd (no explicit-matrix output), plus the dump [n][32]
extern "C" int camera_bwd_lab(const float* kps, const long long* cam_idx, int single_idx, const float* extrinsic, int n_ext,
                              const float* intr_init, const float* intr_noise, float intr_scale, int multiplicative,
                              const float* extr_init, const float* extr_noise, float extr_scale, int n_cams, const float* grid_o,
                              float scale_o, const float* grid_d, float scale_d, int gh, int gw, int H, int W, const float* g_o,
                              const float* g_d, float* d_intr_noise, float* d_extr_noise, float* d_grid_o, float* d_grid_d,
                              float* workspace, float* dump, int n, void* stream) {
    // n_ext carries lab flags (no explicit matrices here): 1 = no LDS stage and no barrier (global atomics per ray), 2 = no memsets
    // before the kernel (the sums are then garbage; the per-ray dump is what is compared), 4 = no finish kernel after it
    const int flags = n_ext;
    n_ext = 0;
    const CamArgs a = make_args(kps, cam_idx, single_idx, extrinsic, n_ext, intr_init, intr_noise, intr_scale, multiplicative, extr_init,
                                extr_noise, extr_scale, n_cams, grid_o, scale_o, grid_d, scale_d, gh, gw, H, W, n);
    const int rc = check_args(a);
    SCN_RETURN_IF(rc != 0, rc);
    SCN_RETURN_IF(!workspace || !dump || extrinsic, SCN_EINVAL);
    hipStream_t st = (hipStream_t)stream;
    const int slots = n_cams;
    if (!(flags & 2)) {
        SCN_HIP(hipMemsetAsync(workspace, 0, sizeof(float) * (4 + 12 * (size_t)slots), st));
        if (d_grid_o) SCN_HIP(hipMemsetAsync(d_grid_o, 0, sizeof(float) * 3 * (size_t)gh * gw, st));
        if (d_grid_d && d_grid_d != d_grid_o) SCN_HIP(hipMemsetAsync(d_grid_d, 0, sizeof(float) * 3 * (size_t)gh * gw, st));
        if (d_extr_noise) SCN_HIP(hipMemsetAsync(d_extr_noise, 0, sizeof(float) * 9 * (size_t)n_cams, st));
    }
    const int lds_slots = (slots > 1024 || (flags & 1)) ? 0 : slots;
    hipLaunchKernelGGL(camera_rays_bwd_dump_kernel, dim3(scn_ceil_div(n, 256)), dim3(256), (size_t)lds_slots * 12 * sizeof(float), st, a,
                       g_o, g_d, workspace, d_grid_o, d_grid_d, lds_slots, dump);
    if (!(flags & 4))
        hipLaunchKernelGGL(camera_finish_kernel, dim3(scn_ceil_div(slots, 64)), dim3(64), 0, st, a, workspace, d_intr_noise, d_extr_noise,
                           (float*)nullptr);
    return scn_launch_status();
}
