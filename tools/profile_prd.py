"""(GPU box) torch profile of one projected-ray-distance term: launches and host time per call (python tools/profile_prd.py)."""
import sys, types, time, torch
sys.path.insert(0, '.')
import bench
from scnerf_amd import synthetic as synth
from scnerf_amd.get_rays import get_rays_kps_use_camera
from scnerf_amd.ray_dist_loss import proj_ray_dist_loss_single
dev = torch.device('cuda:0')
w = bench.build_world(dev, 0, 4096)
cam = w['cam']
N_CAMS, IMG_H, IMG_W = bench.N_CAMS, bench.IMG_H, bench.IMG_W
for t_ in cam.parameters(): pass
E = cam.get_extrinsic().detach().cpu(); K = cam.get_intrinsic().detach().cpu()
k0, k1 = synth.matched_keypoints(IMG_H, IMG_W, K, E[0], E[1], 1024, seed=8)
k0, k1 = k0.to(dev), k1.to(dev)
args = types.SimpleNamespace(proj_ray_dist_threshold=5.0)
i_map = torch.arange(N_CAMS).numpy()
for name in ('intrinsics_noise','extrinsics_noise','ray_o_noise','ray_d_noise'):
    getattr(cam, name).requires_grad_(True)
def prd_step():
    for p in cam.parameters(): p.grad = None
    r0 = get_rays_kps_use_camera(H=IMG_H, W=IMG_W, camera_model=cam, idx_in_camera_param=0, kps_list=k0)
    r1 = get_rays_kps_use_camera(H=IMG_H, W=IMG_W, camera_model=cam, idx_in_camera_param=1, kps_list=k1)
    loss, _ = proj_ray_dist_loss_single(kps0_list=k0, kps1_list=k1, img_idx0=0, img_idx1=1, rays0=r0, rays1=r1, mode="train", device=dev, H=IMG_H, W=IMG_W, args=args, camera_model=cam, method="NeRF", i_map=i_map)
    loss.backward()
for _ in range(5): prd_step()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): prd_step()
torch.cuda.synchronize(); print("ms per call", (time.perf_counter()-t0)/20*1e3)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): prd_step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=35, max_name_column_width=50))
