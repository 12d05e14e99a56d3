"""(GPU box) Which calls of a step make the host wait for the device?  One step of bench.py's --config k under
torch.cuda.set_sync_debug_mode("warn"): every synchronising torch call (.item(), blocking copies, nonzero ...) is reported
with the python line it came from.    python tools/sync_debug.py [--config 3] [--prd-sync]"""
import argparse
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                    # noqa: E402
from scnerf_amd.parallel import FlatGradAllReduce               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--prd-sync", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.config == 4:
        from tools import bench_nerfpp
        step, _, _ = bench_nerfpp.build(2048, device=dev)
        w = None
    else:
        w = bench.build_world(dev, 0, 4096)
    if a.config == 4:
        pass
    elif a.config == 1:
        red = FlatGradAllReduce([w["net_c"], w["net_f"]], 1)
        step = bench.fixed_camera_step(w, red)
    else:
        red = FlatGradAllReduce([w["net_c"], w["net_f"], w["cam"]], 1)
        step = bench.learnable_camera_step(w, red, prd=(a.config == 3), prd_sync=a.prd_sync)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        step()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    print("config %d: %d synchronising call(s) in one step" % (a.config, len(seen)))
    for wmsg in seen:
        print("  %s:%d  %s" % (os.path.relpath(wmsg.filename), wmsg.lineno, str(wmsg.message)[:160]))


if __name__ == "__main__":
    main()
