"""(GPU box, lab) one of this library's heavy kernels in a tight loop for SECONDS -- the HOST process of the guest-write experiments
(tools/ubench/guest_write_lab guest ... or tools/guest_probe.py --external in another process).
    SCNERF_HIP_LIB=<variant.so> python tools/host_loop.py dgrad 20"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scnerf_amd import mlp_layout as ML, ops, synthetic as synth  # noqa: E402


def main():
    what, seconds = sys.argv[1], float(sys.argv[2])
    P = 4096 * 192
    p = synth.network_params(seed=0)
    flat = torch.cat([p[n].reshape(-1) for n, _ in ML.PARAM_SHAPES]).cuda()
    wf, wb, rw = ops.pack_weights(flat, "fwd"), ops.pack_weights(flat, "bwd"), ops.pack_resident(flat)
    pts = torch.rand(P, 3, device="cuda") * 2 - 1
    vd = torch.nn.functional.normalize(torch.randn(4096, 3, device="cuda"), dim=-1)
    save = ops.save_workspace(P, "cuda")
    d_raw = torch.randn(P, 4, device="cuda") * 1e-3
    mx = ops.ChunkMaxima(P, "cuda")
    ops.mlp_fwd(pts, vd, 192, wf, save, planes=rw, maxima=mx)
    t_end = time.time() + seconds
    n = 0
    while time.time() < t_end:
        for _ in range(8):
            if what == "dgrad":
                ops.mlp_bwd(d_raw, pts, vd, 192, wb, save, planes=rw, maxima=mx)
            else:
                ops.mlp_fwd(pts, vd, 192, wf, save, planes=rw, maxima=mx)
            n += 1
        torch.cuda.synchronize()
    print("host loop:", what, n, "launches")


if __name__ == "__main__":
    main()
