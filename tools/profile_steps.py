"""(GPU box) torch profile of the bench's training steps: launches and host time per step, by op
(python tools/profile_steps.py [camera]); `python tools/profile_steps.py pmc` runs PMC_STEPS (default 3) plain steps and
nothing else -- the workload of the whole-step HBM-traffic counters (tools/collect_profiles.sh: FETCH_SIZE / WRITE_SIZE
summed over every dispatch of the process / PMC_STEPS; the few MB of set-up kernels are in the sum)."""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                    # noqa: E402
from scnerf_amd.parallel import FlatGradAllReduce                               # noqa: E402
from torch.profiler import profile, ProfilerActivity                           # noqa: E402

dev = torch.device("cuda:0")
w = bench.build_world(dev, 0, int(os.environ.get("PMC_RAYS", "4096")))
if len(sys.argv) > 1 and sys.argv[1] == "camera":
    for name in ("intrinsics_noise", "extrinsics_noise", "ray_o_noise", "ray_d_noise"):
        getattr(w["cam"], name).requires_grad_(True)
    red = FlatGradAllReduce([w["net_c"], w["net_f"], w["cam"]], 1)
    step = bench.learnable_camera_step(w, red)
else:
    red = FlatGradAllReduce([w["net_c"], w["net_f"]], 1)
    step = bench.fixed_camera_step(w, red)
if len(sys.argv) > 1 and sys.argv[1] == "pmc":
    for _ in range(int(os.environ.get("PMC_STEPS", "3"))):
        step()
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=60))
