"""(GPU box) Trains the coarse + fine network on the procedural scene with the exact-fp32 MFMA arithmetic (the numerical
yardstick, NOT the arithmetic under test) and writes their state dicts as a golden fixture: the weight distribution a converged
NeRF has (heavy-tailed rows, dead units, large biases) instead of xavier's -- tests/trained_weights.py hands it to the parity
tests of the resident arithmetic (reference initialisation vs. trained: NeRF/run_nerf_helpers.py:13-21, :105-128).

    python tools/train_golden_weights.py --steps 5000 --out gpurun_out/r06/trained_nerf.npz      # then: cp -> tests/golden/
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import psnr_trajectory as T
    t0 = time.time()
    curve, nets = T.run_gpu(a.steps, 500, "fp32", return_nets=True)
    out = {}
    stats = {}
    for tag, net in zip(("coarse", "fine"), nets):
        for k, v in net.state_dict().items():
            out["%s/%s" % (tag, k)] = v.detach().cpu().numpy().astype(np.float32)
        for l in range(8):
            W = net.state_dict()["pts_linears.%d.weight" % l].detach().cpu()
            r1 = W.abs().sum(1)
            stats["%s/pts_linears.%d" % (tag, l)] = {
                "row_1norm_max_over_median": float(r1.max() / r1.median()), "abs_max": float(W.abs().max()),
                "abs_median": float(W.abs().median()), "bias_abs_max": float(net.state_dict()["pts_linears.%d.bias" % l].abs().max())}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    np.savez_compressed(a.out, **out)
    meta = {"steps": a.steps, "n_rand": T.N_RAND, "samples": [T.S_C, T.S_F], "lr": T.LR, "arithmetic": "fp32 (v_mfma_f32_32x32x2_f32)",
            "scene": "procedural (scnerf_amd/synthetic.py)", "curve": curve, "seconds": time.time() - t0, "weight_stats": stats}
    with open(os.path.splitext(a.out)[0] + ".json", "w") as fh:
        json.dump(meta, fh, indent=1)
    print(json.dumps({"final_psnr": curve[-1]["psnr"], "seconds": meta["seconds"], "bytes": os.path.getsize(a.out)}))


if __name__ == "__main__":
    main()
