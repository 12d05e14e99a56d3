/* scnerf_hip.h -- C ABI of libscnerf_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the per-ray render path + camera ray generator of SCNeRF.
 * The reference has no FFI on this path (it is torch tensor code); its only
 * native boundary, the unused torchsearchsorted extension, takes caller-allocated outputs
 * and raw tensors (NeRF/torchsearchsorted/src/cuda/searchsorted_cuda_wrapper.cpp:9-16).
 * This ABI keeps that convention:
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's allocator);
 *   - nothing is allocated, freed or synchronised inside; work is enqueued on `stream`
 *     (a hipStream_t passed as void*);
 *   - the return value is 0 on success, otherwise a hipError_t / a negative argument error;
 *   - fp32 everywhere, int64 for sample indices, row-major contiguous unless a stride is named.
 *
 * Each entry cites the reference code it replaces (paths relative to /root/reference).
 */
#ifndef SCNERF_HIP_H
#define SCNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCNERF_ABI_VERSION 4

int scnerf_abi_version(void);

/* ------------------------------------------------------------------ sampling --------- */

/* Batched binary search, out[r, j] = number of a[r, :] elements <= v[r, j] (side 'right') or
 * < v[r, j] (side_left != 0); int64 output.  Replaces torch.searchsorted(cdf, u, right=True)
 * (NeRF/render.py:444) and the function of the vendored extension
 * (NeRF/torchsearchsorted/src/cuda/searchsorted_cuda_kernel.cu:85-107, wrapper .cpp:9-16).
 * a: [nrow_a, na] with nrow_a in {1, nrow}; v: [nrow_v, nv] with nrow_v in {1, nrow}. */
int scnerf_searchsorted(const float* a, const float* v, int64_t* out, int nrow, int nrow_a,
                        int nrow_v, int na, int nv, int side_left, void* stream);

/* Inverse-CDF sampling, NeRF/render.py:417-460 (sample_pdf) with the uniform variates given:
 * bins [n, nb], weights [n, nb-1], u [n, ns] (u_row_stride = ns) or one shared row
 * (u_row_stride = 0, the det=True linspace).  Outputs: samples [n, ns]; optional (may be
 * NULL) inds int64 [n, ns] (the searchsorted result of :444) and cdf [n, nb]. */
int scnerf_sample_pdf(const float* bins, const float* weights, const float* u, int u_row_stride,
                      float* samples, int64_t* inds, float* cdf, int n, int nb, int ns,
                      void* stream);

/* The random numbers of one render_rays call in ONE launch: the stratified jitter t_rand [n_t_rand] in [0, 1)
 * (NeRF/render.py:252-257), the inverse-cdf variates u [n_u] in [0, 1) (:425-429) and the density noise of the coarse and
 * the fine stage, standard normal x raw_noise_std (:329-330, called at :262 and :285).  Any pointer may be NULL (that draw
 * is skipped); non-NULL pointers are 16-byte aligned.  Philox4x32-10 keyed by `seed`, counter = (element / 4, stream, call):
 * a pure function of its arguments, replacing four torch.rand / torch.randn launches and their scaling passes. */
int scnerf_render_randoms(unsigned long long seed, unsigned long long call, float* t_rand, long long n_t_rand,
                          float* u, long long n_u, float* noise_c, long long n_noise_c, float* noise_f,
                          long long n_noise_f, float raw_noise_std, void* stream);

/* Stratified coarse samples + points, NeRF/render.py:235-259.
 * rays [n, ray_stride] = [o(3) d(3) near far ...]; t_vals [s] = linspace(0,1,s) (host made);
 * t_rand [n, s] in [0,1) or NULL (perturb == 0).  Outputs z [n, s], pts [n, s, 3]. */
int scnerf_coarse_sample(const float* rays, int ray_stride, const float* t_vals,
                         const float* t_rand, float* z, float* pts, int n, int s, int lindisp,
                         void* stream);

/* Hierarchical stage between the two networks, NeRF/render.py:268-277:
 * z_mid, sample_pdf(z_mid, w[1:-1], sf), sort(cat(z_c, z_samples)), pts = o + d*z.
 * z_c, w_c [n, sc]; u as in scnerf_sample_pdf.  Outputs z_f [n, sc+sf] (sorted),
 * pts_f [n, sc+sf, 3], z_samples [n, sf], z_std [n] (population std of z_samples, :294);
 * optional inds int64 [n, sf] and cdf [n, sc-1]. */
int scnerf_fine_sample(const float* rays, int ray_stride, const float* z_c, const float* w_c,
                       const float* u, int u_row_stride, float* z_f, float* pts_f,
                       float* z_samples, float* z_std, int64_t* inds, float* cdf, int n, int sc,
                       int sf, void* stream);

/* ------------------------------------------------------------------ camera rays ------ */

/* get_rays_kps_use_camera / get_rays_full_image_use_camera (NeRF/get_rays.py:26-148) fused with the
 * camera model's parameter algebra (model/camera_model.py:24-46, :166-190; camera_utils.py:78-133,
 * :191-195).  kps [n,2] float (x, y) or NULL = every pixel of the H x W image in row-major order.
 * Pose source: explicit `extrinsic` [n_ext,4,4] (n_ext = 1 shared, or n per ray) when non-NULL, else
 * the learnable cameras, indexed per ray by cam_idx [n] (int64) or by `single_idx` when cam_idx is
 * NULL.  intr_* [4] = (fx, fy, cx, cy) initial value and residual ("noise"); extr_* [n_cams, 9] =
 * 6-D rotation + translation; grid_o / grid_d [gh, gw, 3] ray-origin / ray-direction noise grids
 * (either may be NULL = the model has no such attribute); rays_d is renormalised only when grid_d is
 * present (get_rays.py:140-146).  Outputs rays_o, rays_d [n,3]. */
int scnerf_camera_rays_fwd(const float* kps, const long long* cam_idx, int single_idx,
                           const float* extrinsic, int n_ext, const float* intr_init,
                           const float* intr_noise, float intr_scale, int multiplicative,
                           const float* extr_init, const float* extr_noise, float extr_scale,
                           int n_cams, const float* grid_o, float scale_o, const float* grid_d,
                           float scale_d, int gh, int gw, int H, int W, float* rays_o, float* rays_d,
                           int n, void* stream);

/* Gradient of the above w.r.t. the learnable tensors: d_intr_noise [4], d_extr_noise [n_cams,9],
 * d_grid_o / d_grid_d [gh,gw,3], d_extrinsic [n_ext,4,4] (each may be NULL).  g_o, g_d [n,3] may be
 * NULL (= zero).  workspace: scnerf_camera_bwd_workspace_floats(n_cams or n_ext) floats. */
long long scnerf_camera_bwd_workspace_floats(int n_slots);
int scnerf_camera_rays_bwd(const float* kps, const long long* cam_idx, int single_idx,
                           const float* extrinsic, int n_ext, const float* intr_init,
                           const float* intr_noise, float intr_scale, int multiplicative,
                           const float* extr_init, const float* extr_noise, float extr_scale,
                           int n_cams, const float* grid_o, float scale_o, const float* grid_d,
                           float scale_d, int gh, int gw, int H, int W, const float* g_o,
                           const float* g_d, float* d_intr_noise, float* d_extr_noise, float* d_grid_o,
                           float* d_grid_d, float* d_extrinsic, float* workspace, int n, void* stream);

/* CameraModel.get_intrinsic() / get_extrinsic() (model/camera_model.py:160-192 with model/camera_utils.py:78-133,
 * :191-195) as one launch each way: K [4,4] = identity with fx, fy, cx, cy = intr_init + intr_noise * intr_scale (x
 * intr_init when multiplicative); E [n_cams,4,4] = [Gram-Schmidt rotation of the first six of extr_init + extr_scale *
 * extr_noise | translation = the last three; 0 0 0 1].  Backward: g_K [4,4] / g_E [n_cams,4,4] (each may be NULL = zero)
 * -> d_intr_noise [4], d_extr_noise [n_cams,9] (each may be NULL). */
int scnerf_camera_matrices_fwd(const float* intr_init, const float* intr_noise, float intr_scale, int multiplicative,
                               const float* extr_init, const float* extr_noise, float extr_scale, int n_cams,
                               float* K, float* E, void* stream);
int scnerf_camera_matrices_bwd(const float* intr_init, float intr_scale, int multiplicative, const float* extr_init,
                               const float* extr_noise, float extr_scale, int n_cams, const float* g_K,
                               const float* g_E, float* d_intr_noise, float* d_extr_noise, void* stream);

/* get_rays_kps_no_camera / get_rays_full_image_no_camera (NeRF/get_rays.py:5-23, :75-90): pinhole
 * rays at the truncated pixel coordinates kps [n, kps_stride>=2] (or every pixel when NULL) through
 * the fixed pose c2w [4,4]. */
int scnerf_pinhole_rays(const float* kps, int kps_stride, const float* c2w, float focal, int H, int W,
                        float* rays_o, float* rays_d, int n, void* stream);

/* ndc_rays / ndc_rays_camera (NeRF/render.py:357-396); focal_xy = device pointer to (fx, fy). */
int scnerf_ndc_fwd(int H, int W, const float* focal_xy, float near, const float* rays_o,
                   const float* rays_d, float* ndc_o, float* ndc_d, int n, void* stream);
int scnerf_ndc_bwd(int H, int W, const float* focal_xy, float near, const float* rays_o,
                   const float* rays_d, const float* g_ndc_o, const float* g_ndc_d, float* g_rays_o,
                   float* g_rays_d, float* g_focal_xy, int n, void* stream);

/* The rest of render() between the ray source and batchify_rays (NeRF/render.py:105-128) in one launch:
 * ray_batch [n, cols] = [o', d', near, far (, viewdir)] with cols = 8 or 11; viewdir = rays_d / |rays_d| of the
 * un-warped ray; (o', d') = the NDC warp (focal_xy: 2 device floats, ndc_near as scnerf_ndc_fwd) or, with
 * focal_xy == NULL (ndc = False), the rays themselves.  _bwd: gradients to rays_o, rays_d and (optional) the two
 * focal lengths; the near / far columns carry none. */
int scnerf_pack_rays_fwd(int H, int W, const float* focal_xy, float ndc_near, const float* rays_o, const float* rays_d,
                         float near, float far, int cols, float* ray_batch, int n, void* stream);
int scnerf_pack_rays_bwd(int H, int W, const float* focal_xy, float ndc_near, const float* rays_o, const float* rays_d,
                         int cols, const float* g_ray_batch, float* g_rays_o, float* g_rays_d, float* g_focal_xy, int n,
                         void* stream);

/* CameraModel.get_ray_o_noise / get_ray_d_noise (model/camera_model.py:24-46): bilinear
 * (align_corners=False) upsampling of a [gh,gw,3] grid to [H*W,3], times scale; and its gradient. */
int scnerf_upsample_grid_fwd(const float* grid, float scale, int gh, int gw, int H, int W, float* out,
                             void* stream);
int scnerf_upsample_grid_bwd(const float* g_out, float scale, int gh, int gw, int H, int W,
                             float* d_grid, void* stream);

/* ------------------------------------------------------------------ compositing ------ */

/* raw2outputs, NeRF/render.py:302-355.  raw [n, s, 4]; z [n, s]; rays [n, ray_stride] (columns
 * 3:6 = rays_d, only its norm is used); noise [n, s] = the already scaled density noise of
 * :329-336, or NULL.  Outputs rgb_map [n,3], disp_map [n], acc_map [n]; optional depth_map [n]
 * and weights [n, s] (needed by the hierarchical sampler). */
int scnerf_composite_fwd(const float* raw, const float* z, const float* rays, int ray_stride,
                         const float* noise, int white_bkgd, float* rgb_map, float* disp_map,
                         float* acc_map, float* depth_map, float* weights, int n, int s,
                         void* stream);

/* Gradient of raw2outputs.  g_rgb [n,3], g_disp, g_acc, g_depth [n] (each may be NULL = zero);
 * g_raw_in [n,s,4] optional gradient arriving directly at raw.  Outputs d_raw [n,s,4] and
 * d_rays_d [n,3] (through |rays_d| of :325; may be NULL).  z_vals get no gradient (the
 * reference detaches z_samples, :274, and near/far are constants in run_nerf.py). */
int scnerf_composite_bwd(const float* raw, const float* z, const float* rays, int ray_stride,
                         const float* noise, int white_bkgd, const float* g_rgb,
                         const float* g_disp, const float* g_acc, const float* g_depth,
                         const float* g_raw_in, float* d_raw, float* d_rays_d, int n, int s,
                         void* stream);

/* d ray_batch from the per-sample point / view-direction gradients of scnerf_mlp_bwd
 * (pts = o + d z, NeRF/render.py:259,277; viewdirs broadcast, create_nerf.py:25):
 * d_rays[:,0:3] = sum_s d_pts, [:,3:6] = sum_s d_pts z + extra_d, [:,8:11] = sum_s d_views,
 * [:,6:8] = 0; accumulate != 0 adds into d_rays instead (second network). */
int scnerf_ray_reduce(const float* d_pts, const float* d_views, const float* z,
                      const float* extra_d, float* d_rays, int ray_stride, int accumulate, int n,
                      int s, void* stream);

/* ------------------------------------------------------------------ NeRF MLP --------- */

/* dst[i] = idx[i] >= 0 ? src[idx[i]] : 0 -- packs the flat parameter buffer of one NeRF
 * (reference parameter order, NeRF/run_nerf_helpers.py:88-103) into the MFMA streaming
 * order described by the index tables of scnerf_amd/mlp_layout.py. */
int scnerf_gather_f32(const float* src, const int* idx, float* dst, long long n, void* stream);

/* Layout constants compiled into the kernels (checked against mlp_layout.py at load) for the network
 * variant pt_dims (see scnerf_mlp_fwd); n >= 22. */
int scnerf_mlp_layout_info(int pt_dims, int* out, int n);

/* Fused positional encoding + NeRF.forward for the standard network (D=8, W=256, skips=[4],
 * use_viewdirs, multires 10/4): replaces run_network + Embedder + NeRF.forward
 * (NeRF/create_nerf.py:18-32, NeRF/run_nerf_helpers.py:24-72, :105-128) and, with the parameters
 * renamed, NeRF++'s Embedder + MLPNet.forward before its output non-linearities
 * (nerfplusplus/nerf_network.py:41-60, :117-142).  pt_dims = 3: points (x, y, z), 63 encoded columns
 * -- the SCNeRF networks and NeRF++'s foreground net; pt_dims = 4: points (x, y, z, 1/r), 84 encoded
 * columns -- NeRF++'s background net (nerfplusplus/ddp_model.py:62-71).
 * pts [n_samples, pt_dims]; viewdirs row r at viewdirs + r * vd_stride (3 for a packed [n_rays, 3]
 * tensor, 11 with viewdirs = ray_batch + 8), ray(p) = p / samples_per_ray;
 * wpacked = forward packed buffer (scnerf_gather_f32 with mlp_layout.forward_index());
 * raw [n_samples, 4] = (rgb logits, sigma).  save: NULL (inference) or the activation
 * workspace of mlp_layout.SAVE_FLOATS_PER_SAMPLE * n_samples floats (training). */
int scnerf_mlp_fwd(int pt_dims, const float* pts, const float* viewdirs, int vd_stride, int samples_per_ray,
                   const float* wpacked, float* raw, float* save, long long n_samples,
                   void* stream);

/* Workspace sizes (floats) for n_samples samples: activations saved by the training forward
 * (rows + ReLU bit masks) and the per-layer output gradients written by scnerf_mlp_bwd. */
long long scnerf_mlp_save_floats(int pt_dims, long long n_samples);
long long scnerf_mlp_grad_floats(long long n_samples);

/* The coarse stage of render_rays (NeRF/render.py:235-262) as ONE launch: scnerf_coarse_sample + scnerf_mlp_fwd
 * (pt_dims = 3, samples_per_ray = 64, view directions = columns 8..10 of the ray batch) + scnerf_composite_fwd, with
 * the stratified depths computed in the network kernel's prologue and the compositing in its epilogue (the two
 * rays of a workgroup meet in LDS).  Same arguments and bit-identical outputs as those three calls; n_samples must
 * be 64 (anything else: SCN_ENOSUP, use the three calls); save == NULL selects the inference instantiation. */
int scnerf_coarse_stage_fwd(const float* rays, int ray_stride, const float* t_vals, const float* t_rand, int lindisp,
                            const float* wpacked, float* save, const float* noise, int white_bkgd, float* z,
                            float* pts, float* raw, float* rgb_map, float* disp_map, float* acc_map,
                            float* depth_map, float* weights, int n_rays, int n_samples, void* stream);

/* Data-gradient chain of the fused network (what autograd derives from NeRF.forward,
 * Embedder and run_network: NeRF/run_nerf_helpers.py:105-128, :24-72, create_nerf.py:18-32).
 * d_raw [n_samples, 4]; wpacked_bwd = backward packed buffer (mlp_layout.backward_index());
 * save = workspace filled by scnerf_mlp_fwd on the same inputs.  Outputs: grads workspace
 * (dZ of the 8 trunk layers, d feature, dZ of the views layer: row-major, consumed by
 * scnerf_wgrad), d_pts [n_samples, pt_dims], d_views [n_samples, 3] (per sample, not yet summed
 * per ray). */
int scnerf_mlp_bwd(int pt_dims, const float* d_raw, const float* pts, const float* viewdirs, int vd_stride,
                   int samples_per_ray, const float* wpacked_bwd, const float* save,
                   float* grads, float* d_pts, float* d_views, long long n_samples, void* stream);

/* Weight / bias gradient of one nn.Linear as a GEMM reduced over the samples (what autograd
 * derives for the layers of NeRF/run_nerf_helpers.py:88-128):
 *   dW[n * ldo + col0 + k] = sum_p dz[p][n] * x[p][k]     n < n_out, k < k_out
 *   db[n] = sum_p dz[p][n]                                  (db may be NULL)
 * Each operand is either row-major [n_samples][ld] (*_tiled = 0; columns < n_load / k_load, multiples
 * of 4 and <= 256, are read with 16-byte loads, so the pointer must be 16-byte aligned and ld a multiple
 * of 4) or a TILE-NATIVE section of width ld as written by scnerf_mlp_fwd / scnerf_mlp_bwd (*_tiled = 1:
 * per 32 samples a block [ld/32][4][64 lanes][4]; ld must equal n_load / k_load and be 64, 128 or 256).
 * The samples are split into n_chunks workgroups whose partial results are summed in a fixed order;
 * `workspace` holds scnerf_wgrad_workspace_floats(n_load, k_load, n_chunks) floats. */
long long scnerf_wgrad_workspace_floats(int n_load, int k_load, int n_chunks);
int scnerf_wgrad(const float* dz, int lda, int n_load, int n_out, int dz_tiled, const float* x,
                 int ldb, int k_load, int k_out, int x_tiled, long long n_samples, int n_chunks,
                 float* workspace, float* dW, int ldo, int col0, float* db, void* stream);

/* Gradient of a one-row linear layer (alpha_linear, NeRF/run_nerf_helpers.py:100,115):
 * dv[k] = sum_p vec[p * vec_stride] * x[p][k] (k < 256) for a tile-native x of width 256, and
 * *dvsum = sum_p vec[p * vec_stride] (may be NULL).  workspace: >= 257 * n_chunks floats. */
int scnerf_vecmat(const float* x_tiled256, const float* vec, int vec_stride, long long n_samples,
                  int n_chunks, float* workspace, float* dv, float* dvsum, void* stream);

/* All weight and bias gradients of one network (variant pt_dims, see scnerf_mlp_fwd), written into a
 * flat gradient buffer in the reference NeRF's parameter order (scnerf_nerf_param_count(pt_dims)
 * floats; NeRF/run_nerf_helpers.py:88-103): 12 scnerf_wgrad calls over the workspaces of
 * scnerf_mlp_fwd (save) and scnerf_mlp_bwd (grads) and d_raw [n_samples, 4].  workspace:
 * scnerf_nerf_wgrad_workspace_floats(n_chunks) floats. */
int scnerf_nerf_param_count(int pt_dims);
/* Arithmetic of the 256 x 256 weight-gradient GEMMs (87 % of the weight-gradient FLOPs) and of the narrow ones with a
 * tile-native dZ.  mode 1 (the default): three fp16 products per product -- every fp32 operand scaled by one power of two
 * per operand and workgroup chunk and cut into two fp16 numbers, fp32 accumulation; the error against fp64 equals the
 * exact-fp32 kernel's -- wherever the operands' chunk maxima are known (scnerf_nerf_wgrad_h3); mode 0, and where no
 * maxima were left: v_mfma_f32_32x32x2_f32.  Any other value only queries.  Returns the mode in force.  Environment
 * preset: SCNERF_WGRAD_ARITHMETIC=fp32 | half.  (No reference counterpart: torch.autograd computes these GEMMs with
 * whatever sgemm the build links.) */
int scnerf_wgrad_arithmetic(int mode);
long long scnerf_nerf_wgrad_workspace_floats(int n_chunks);
/* scnerf_nerf_wgrad with the eight 256 x 256 GEMMs on THREE fp16 products (csrc/wgrad256_half.h) when the chunk maxima
 * of their operands are given -- amax_x / amax_z [8][scnerf_wgrad256_chunks(n_chunks)], left by scnerf_mlp_fwd_h3 /
 * scnerf_coarse_stage_fwd_h3 and scnerf_mlp_bwd_h3 for that chunk count -- and the arithmetic in force is 1 (the default); otherwise as
 * scnerf_nerf_wgrad.  With `scales` (the table of scnerf_h3_pack) as well, the narrow GEMMs with a tile-native dZ
 * (256 x 64 / 128 of the encoded-point layers, 128 x 256 of the views layer) run on three fp16 products too
 * (csrc/wgrad_half_narrow.h): amax_z then has 12 rows -- 8: dZ of the views layer, 9: dZ of layer 0, 10: max(1, |point|),
 * 11: max(1, |direction|), all left by scnerf_mlp_bwd_h3 -- and the feature is bounded through amax_x row 7 and the
 * table.
 * ev_before / ev_after: NULL, or two hipEvent_t (created by the caller) recorded on the stream right before and after
 * the call's one launch of the eight 256 x 256 GEMMs (bench.py's per-kernel timing; arguments of this call, nothing is
 * kept between calls).
 * scnerf_wgrad_chunk_samples: the samples per workgroup chunk both sides use.
 * scnerf_wgrad256_half: one such GEMM with given maxima [n_chunks] (accuracy tests); workspace n_chunks * (65536 + 256). */
long long scnerf_wgrad_chunk_samples(long long n_samples, int n_chunks);
/* chunks the eight 256 x 256 GEMMs of scnerf_nerf_wgrad[_h3] split the samples into when the call is given n_chunks (an
 * eighth: 32 x 8 jobs = one workgroup per CU): the chunk count of amax_x / amax_z. */
int scnerf_wgrad256_chunks(int n_chunks);
int scnerf_nerf_wgrad_h3(int pt_dims, const float* save, const float* grads, const float* d_raw,
                         long long n_samples, int n_chunks, float* workspace, float* flat_grad,
                         int accumulate, const float* amax_x, const float* amax_z, const float* scales,
                         void* ev_before, void* ev_after, void* stream);
int scnerf_wgrad256_half(const float* dz_tiled, const float* x_tiled, long long n_samples, int n_chunks,
                         float* workspace, float* dW, float* db, const float* amax_dz, const float* amax_x,
                         void* stream);
/* One of the narrow weight-gradient GEMMs on three fp16 products (csrc/wgrad_half_narrow.h; accuracy tests): dZ tile-native
 * of width n_load (256 / 128), X row-major [n_samples][k_load] (k_load 64 / 128, x_tiled 0) or tile-native (k_load 256,
 * x_tiled 1); amax_dz / amax_x [n_coarse]: the operands' maxima per chunk of coarse_chunk samples; workspace
 * scnerf_wgrad_workspace_floats(n_load, k_load, n_chunks); dW [n_load][k_out], db [n_load] or NULL. */
int scnerf_wgrad_half_narrow(const float* dz_tiled, int n_load, const float* x, int k_load, int k_out, int x_tiled,
                             long long n_samples, int n_chunks, float* workspace, float* dW, float* db,
                             const float* amax_dz, const float* amax_x, int n_coarse, long long coarse_chunk,
                             void* stream);
 /* accumulate != 0: flat_grad += the gradients (autograd's accumulation into an attached flat .grad
 * buffer without 48 separate add kernels); 0: overwrite. */
int scnerf_nerf_wgrad(int pt_dims, const float* save, const float* grads, const float* d_raw,
                      long long n_samples, int n_chunks, float* workspace, float* flat_grad,
                      int accumulate, void* stream);

/* ------------------------------------------------------------------ resident arithmetic ---- */
/* The whole network in ONE launch on three fp16 products per product with register-resident activations
 * (csrc/mlp_h3.h): what scnerf_mlp_fwd / scnerf_coarse_stage_fwd / scnerf_mlp_bwd compute (run_network + Embedder +
 * NeRF.forward, NeRF/create_nerf.py:18-32, NeRF/run_nerf_helpers.py:24-72, :105-128, and for the coarse stage
 * NeRF/render.py:235-262), to the same workspaces.
 *
 * scnerf_h3_pack: flat parameters (reference registration order) -> the two fragment streams (fp16 planes of
 * weight x the layer's power-of-two scale, in the order a wave consumes them) and the scale table [12][8] floats
 * (per layer: Sw, 1 / Sw, largest row 1-norm A, largest |bias| B, largest column 1-norm A'; scnerf_h3_scale_floats()
 * floats: the table followed by the scale pass's scratch); once per optimizer
 * step.  jobs [12][4] ints (weight offset, rows, columns, bias offset), idx_* [frags * 512] ints (flat parameter
 * index or -1), meta_* [frags] bytes (plane | layer << 1): the tables of scnerf_amd/mlp_layout.py (h3_plan,
 * h3_scale_jobs), device pointers.  A direction with frags == 0 is skipped. */
long long scnerf_h3_scale_floats(void);
int scnerf_h3_pack(const float* flat_params, const int* jobs, const int* idx_fwd, const unsigned char* meta_fwd,
                   long long frags_fwd, const int* idx_bwd, const unsigned char* meta_bwd, long long frags_bwd,
                   short* stream_fwd, short* stream_bwd, float* scales, void* stream);
/* wpacked: the packed fp32 buffer of scnerf_gather_f32 (its lane-vector tables: biases, density head); save == NULL:
 * inference.  Arguments otherwise as scnerf_mlp_fwd / scnerf_coarse_stage_fwd. */
/* chunk_amax (or NULL; training): [8][n_chunks] floats (scnerf_mlp_bwd_h3: [12][n_chunks]), zeroed by the caller -- the kernel leaves there, per
 * weight-gradient workgroup chunk of chunk_samples samples (scnerf_wgrad_chunk_samples), the largest |value| of the X
 * operand of each of the eight 256 x 256 weight-gradient GEMMs (scnerf_nerf_wgrad_h3). */
int scnerf_mlp_fwd_h3(int pt_dims, const float* pts, const float* viewdirs, int vd_stride, int samples_per_ray,
                      const float* wpacked, const short* stream_fwd, const float* scales, float* raw, float* save,
                      long long n_samples, float* chunk_amax, int n_chunks, long long chunk_samples, void* stream);
/* scnerf_mlp_bwd in the resident arithmetic; wpacked_bwd: the packed fp32 backward buffer (its density-head table). */
int scnerf_mlp_bwd_h3(int pt_dims, const float* d_raw, const float* pts, const float* viewdirs, int vd_stride,
                      int samples_per_ray, const float* wpacked_bwd, const short* stream_bwd, const float* scales,
                      const float* save, float* grads, float* d_pts, float* d_views, long long n_samples,
                      float* chunk_amax, int n_chunks, long long chunk_samples, void* stream);
int scnerf_coarse_stage_fwd_h3(const float* rays, int ray_stride, const float* t_vals, const float* t_rand,
                               int lindisp, const float* wpacked, const short* stream_fwd, const float* scales,
                               float* save, const float* noise, int white_bkgd, float* z, float* pts, float* raw,
                               float* rgb_map, float* disp_map, float* acc_map, float* depth_map, float* weights,
                               int n_rays, int n_samples, float* chunk_amax, int n_chunks, long long chunk_samples,
                               void* stream);
/* The whole FINE stage of render_rays (NeRF/render.py:269-285) as one launch in the resident arithmetic: scnerf_fine_sample
 * (bins = mid points of z_c, weights w_c[1:-1], inverse cdf at u, merge by rank: the stand-alone kernel's own instructions,
 * csrc/ray_sample.h -- same indices, same depths) in front of the network, raw2outputs (:302-355) behind it.  A workgroup
 * takes whole rays through passes of 128 samples: n_coarse must be 64 and n_importance 64, 128 or 192 (SCN_ENOSUP
 * otherwise: use scnerf_fine_sample + scnerf_mlp_fwd_h3 + scnerf_composite_fwd).  Inputs as scnerf_fine_sample (u
 * [n_rays, n_importance], or one row with u_row_stride 0) and scnerf_mlp_fwd_h3; outputs: z_f [n, 64 + n_importance] and
 * pts_f [n, 64 + n_importance, 3] (what the data-gradient pass reads), z_samples [n, n_importance], z_std [n], optionally
 * inds (int64 [n, n_importance]) and cdf [n, 63]; raw [n, 64 + n_importance, 4]; rgb_map [n,3], disp_map, acc_map [n] and
 * optionally depth_map [n], weights [n, 64 + n_importance]; noise: the density noise [n, 64 + n_importance] or NULL;
 * save / chunk_amax as scnerf_mlp_fwd_h3. */
int scnerf_fine_stage_fwd_h3(const float* rays, int ray_stride, const float* z_c, const float* w_c, const float* u,
                             int u_row_stride, const float* wpacked, const short* stream_fwd, const float* scales,
                             float* save, const float* noise, int white_bkgd, float* z_f, float* pts_f, float* z_samples,
                             float* z_std, long long* inds, float* cdf, float* raw, float* rgb_map, float* disp_map,
                             float* acc_map, float* depth_map, float* weights, int n_rays, int n_coarse, int n_importance,
                             float* chunk_amax, int n_chunks, long long chunk_samples, void* stream);

/* ------------------------------------------------------------------ PRD loss --------- */

/* Projected-ray-distance loss, proj_ray_dist_loss_single (model/ray_dist_loss.py:22-246) after its
 * argument plumbing: kps0/kps1 [m,2] matched key points, rays*_o / rays*_d [m,3] their rays, K [4,4]
 * the intrinsic matrix (negate_fx != 0 flips K[0][0] as the reference does for method "NeRF",
 * :102-105), E2 [2,4,4] the two camera-to-world poses.  Train mode (eval_mode = 0): 0.5 * (mean of the
 * valid image-0 errors + mean of the valid image-1 errors), valid = chirality (t0, t1 > 0) and error
 * < threshold and finite (:213-229); *n_match = matches valid both ways.  Eval mode: invalid errors
 * are replaced by the threshold and averaged over the chirality-valid matches (:231-246).
 * sums6: 6-float scratch kept for the backward.  loss / n_match: device scalars. */
int scnerf_prd_loss_fwd(const float* kps0, const float* kps1, const float* rays0_o, const float* rays0_d,
                        const float* rays1_o, const float* rays1_d, const float* K, const float* E2,
                        float eps, float threshold, int negate_fx, int eval_mode, int m, float* sums6,
                        float* loss, float* n_match, void* stream);

/* Gradient of the train-mode loss w.r.t. the four ray tensors, K (w.r.t. the matrix as passed) and
 * E2; g_loss is a device scalar; workspace36: 36 floats. */
int scnerf_prd_loss_bwd(const float* kps0, const float* kps1, const float* rays0_o, const float* rays0_d,
                        const float* rays1_o, const float* rays1_d, const float* K, const float* E2,
                        float eps, float threshold, int negate_fx, int m, const float* sums6,
                        const float* g_loss, float* g_rays0_o, float* g_rays0_d, float* g_rays1_o,
                        float* g_rays1_d, float* g_K, float* g_E2, float* workspace36, void* stream);

/* Stand-alone positional encoding, Embedder.embed (NeRF/run_nerf_helpers.py:24-55): x [n,d] -> out
 * [n, d (include_input + 2 n_freqs)] = [x | sin(x f0) | cos(x f0) | sin(x f1) | ...]; freqs: n_freqs device
 * floats (the reference's freq_bands).  _bwd: g_x [n,d] from g_out.  (The render path encodes inside the fused
 * network kernels; this is for callers of embed_fn.) */
int scnerf_embed_fwd(const float* x, long long n, int d, const float* freqs, int n_freqs, int include_input,
                     float* out, void* stream);
int scnerf_embed_bwd(const float* x, const float* g_out, long long n, int d, const float* freqs, int n_freqs,
                     int include_input, float* g_x, void* stream);

/* filter_matches_with_gt (model/prd_evaluation.py:189-332): keep[i] = 1 when match i re-projects within
 * `threshold` (the reference uses 1.0, squared pixels) both ways through the GROUND-TRUTH K (negate_fx as above)
 * and E2 = the two ground-truth poses, and both closest points lie in front of their cameras.  keep: m bytes. */
int scnerf_prd_filter(const float* kps0, const float* kps1, const float* rays0_o, const float* rays0_d,
                      const float* rays1_o, const float* rays1_d, const float* K, const float* E2,
                      float eps, float threshold, int negate_fx, int m, unsigned char* keep, void* stream);

/* ------------------------------------------------------------------ NeRF++ ----------- */
/* The per-ray pieces of the NeRF++ path around the fused networks; the foreground
 * network is scnerf_mlp_* with pt_dims = 3, the background one with pt_dims = 4. */

/* intersect_sphere (nerfplusplus/ddp_train_nerf.py:50-68): depth at which each ray leaves the unit
 * sphere.  *outside_flag (optional device int, zeroed by the caller) is set when a ray's closest
 * point to the centre lies outside the sphere -- the reference raises there (:60-64).  _bwd: what
 * autograd derives (gradients to ray_o, ray_d). */
int scnerf_npp_intersect_fwd(const float* ray_o, const float* ray_d, float* far, int* outside_flag, int n,
                             void* stream);
int scnerf_npp_intersect_bwd(const float* ray_o, const float* ray_d, const float* g_far, float* g_ray_o,
                             float* g_ray_d, int n, void* stream);

/* perturb_samples (ddp_train_nerf.py:71-80) with the uniforms t_rand [n,s] supplied by the caller:
 * out = lower + (upper - lower) t, lower / upper the mid points to the neighbours; _bwd: d z. */
int scnerf_npp_perturb_fwd(const float* z, const float* t_rand, float* out, int n, int s, void* stream);
int scnerf_npp_perturb_bwd(const float* g_out, const float* t_rand, float* g_z, int n, int s, void* stream);

/* sample_pdf of NeRF++ (ddp_train_nerf.py:83-132): bins [n,m+1], weights [n,m] (+1e-6, normalised,
 * cumulated with a leading 0), u [n,ns]; upper index = count of cdf[0..m-1] <= u (:113), denominators
 * under 1e-6 -> 1, bin width + 1e-6 (:130).  below_above (int, lower | upper << 16) and t [n,ns] are
 * optional outputs for _bwd, which returns d bins [n,m+1] (the weights are detached by the caller);
 * cdf [n,m+1] (optional) is the cumulated pdf the search ran on (:95-98) -- with below_above what the
 * parity tests compare bit for bit against the reference's own cumsum and comparison count. */
int scnerf_npp_sample_pdf(const float* bins, const float* weights, const float* u, float* samples,
                          int* below_above, float* t, float* cdf, int n, int m, int ns, void* stream);
int scnerf_npp_sample_pdf_bwd(const float* g_samples, const int* below_above, const float* t, float* g_bins,
                              int n, int m, int ns, void* stream);

/* Sample placement of NerfNet.forward (nerfplusplus/ddp_model.py:80-89, :105-114): fg_pts [n,sf,3] =
 * o + z d; bg_pts [n,sb,4] = depth2pts_outside (:16-45) of bg_z in FLIPPED order (the network sees the
 * background far -> near); viewdirs [n,3] = d / |d| (optional); bg_depth_real [n,sb] (optional) = the
 * metric depth of the background samples (:44); sf may be 0.  _bwd: gradients of the two point sets, of the
 * per-sample view directions of both networks and of |d| (scnerf_npp_composite_bwd) summed into d ray_o,
 * d ray_d [n,3]; d fg_z [n,sf] = d_fg_pts . d + g_fg_z_in (optional).  The inverse radii bg_z carry no
 * gradient (they never depend on learnable quantities). */
int scnerf_npp_points_fwd(const float* ray_o, const float* ray_d, const float* fg_z, const float* bg_z,
                          float* fg_pts, float* bg_pts, float* viewdirs, float* bg_depth_real, int n, int sf,
                          int sb, void* stream);
int scnerf_npp_points_bwd(const float* ray_o, const float* ray_d, const float* fg_z, const float* bg_z,
                          const float* d_fg_pts, const float* d_bg_pts, const float* d_views_fg,
                          const float* d_views_bg, const float* d_norm, const float* g_fg_z_in,
                          float* g_ray_o, float* g_ray_d, float* g_fg_z, int n, int sf, int sb, void* stream);

/* Compositing of NerfNet.forward (ddp_model.py:90-143) from the raw network outputs (rgb logits, sigma
 * before abs -- MLPNet's sigmoid / abs, nerf_network.py:133,140, are applied here): foreground intervals
 * scaled by |ray_d| with the last reaching fg_z_max, T = cumprod(1 - alpha + 1e-6), bg_lambda = T behind
 * the last sample; background over the flipped inverse radii with a 1e10 last interval; bg_rgb /
 * bg_depth already scaled by bg_lambda; rgb = fg_rgb + bg_rgb.  raw_bg / bg_weights are in the flipped
 * (network) order, bg_z in the caller's ascending order.  (sb = 1: the single sample is kept; the reference's
 * slicing, :123-124, yields an empty transmittance there and drops the background -- unused: sb >= 2.)
 * _bwd: any incoming gradient may be NULL;
 * returns d raw of both networks, d fg_z, d fg_z_max, d |ray_d|. */
int scnerf_npp_composite_fwd(const float* raw_fg, const float* raw_bg, const float* fg_z, const float* fg_z_max,
                             const float* bg_z, const float* ray_d, float* rgb, float* fg_weights,
                             float* bg_weights, float* fg_rgb, float* fg_depth, float* bg_rgb, float* bg_depth,
                             float* bg_lambda, int n, int sf, int sb, void* stream);
int scnerf_npp_composite_bwd(const float* raw_fg, const float* raw_bg, const float* fg_z, const float* fg_z_max,
                             const float* bg_z, const float* ray_d, const float* g_rgb, const float* g_fg_weights,
                             const float* g_bg_weights, const float* g_fg_rgb, const float* g_fg_depth,
                             const float* g_bg_rgb, const float* g_bg_depth, const float* g_bg_lambda,
                             float* d_raw_fg, float* d_raw_bg, float* d_fg_z, float* d_fg_z_max, float* d_norm,
                             int n, int sf, int sb, void* stream);

/* render_ray_from_camera (nerfplusplus/nerf_sample_ray_split.py:196-257): rays of the
 * row-major pixel indices select [n] (int64) through the CENTRES of those pixels, optional radial
 * distortion dist2 = (k0, k1): p <- (p - c)(1 + r^2 k0 + r^4 k1) + c, r = (p - c) / c per axis (:225-232);
 * K^-1 [u, v, 1] without axis flips; pose = camera `camera_idx` of the model, or the explicit c2w
 * `extrinsic` [4,4] when non-NULL; the ray noise images sampled at the selected pixels; direction
 * renormalised without an epsilon.  Camera-model arguments as scnerf_camera_rays_fwd.  _bwd additionally
 * returns d dist2 [2]; workspace: scnerf_camera_bwd_workspace_floats(n_cams or 1). */
int scnerf_npp_camera_rays_fwd(const long long* select, const float* dist2, int camera_idx,
                               const float* extrinsic, const float* intr_init, const float* intr_noise,
                               float intr_scale, int multiplicative, const float* extr_init,
                               const float* extr_noise, float extr_scale, int n_cams, const float* grid_o,
                               float scale_o, const float* grid_d, float scale_d, int gh, int gw, int H, int W,
                               float* rays_o, float* rays_d, int n, void* stream);
int scnerf_npp_camera_rays_bwd(const long long* select, const float* dist2, int camera_idx,
                               const float* extrinsic, const float* intr_init, const float* intr_noise,
                               float intr_scale, int multiplicative, const float* extr_init,
                               const float* extr_noise, float extr_scale, int n_cams, const float* grid_o,
                               float scale_o, const float* grid_d, float scale_d, int gh, int gw, int H, int W,
                               const float* g_o, const float* g_d, float* d_intr_noise, float* d_extr_noise,
                               float* d_grid_o, float* d_grid_d, float* d_extrinsic, float* d_dist2,
                               float* workspace, int n, void* stream);

/* ------------------------------------------------------------------ optimizer -------- */

/* One Adam step over a flat fp32 segment (f_custom_adam / torch.optim.Adam without amsgrad,
 * NeRF/create_nerf.py:199-254): param, exp_avg, exp_avg_sq updated in place from grad; `step` is the
 * 1-based step count of the segment (bias corrections), weight_decay != 0 adds weight_decay * param
 * to the gradient first (the reference applies it to the trailing ray-noise / distortion tensors
 * only, :219-226, :238-239).  All pointers 16-byte aligned. */
int scnerf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                     long long n, double lr, double beta1, double beta2, double eps,
                     double weight_decay, long long step, void* stream);
/* The same with the weight decay restricted to elements [decay_lo, decay_hi) of the segment (whole tensors at the
 * segment's end: the reference's decayed tail, :219-226, may begin at a tensor boundary that is not 16-byte aligned --
 * e.g. rgb_linear.weight of the fine network while every camera tensor is frozen -- so the segment is not split). */
int scnerf_adam_step_range(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                           long long n, double lr, double beta1, double beta2, double eps,
                           double weight_decay, long long decay_lo, long long decay_hi, long long step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCNERF_HIP_H */
