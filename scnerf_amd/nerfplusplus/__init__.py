"""MI355X-native drop-ins for the NeRF++ half of SCNeRF (/root/reference nerfplusplus/): same module,
class and function names as the reference, work done by the HIP kernels behind include/scnerf_hip.h.

    nerf_network.py          Embedder, MLPNet                    (nerfplusplus/nerf_network.py)
    ddp_model.py             depth2pts_outside, NerfNet, NerfNetWithAutoExpo, remap_name  (ddp_model.py)
    ddp_train_nerf.py        intersect_sphere, perturb_samples, sample_pdf  (the per-ray helpers of the
                             training script, ddp_train_nerf.py:50-132)
    nerf_sample_ray_split.py render_ray_from_camera               (nerf_sample_ray_split.py:196-257)
    create_nerf.py           create_nerf                          (nerfplusplus/create_nerf.py)
"""
