"""(GPU box) 5000 training steps on the procedural scene in BOTH arithmetics from identical initial weights, batches and randoms
(tools/psnr_trajectory.run_gpu): the resident arithmetic (three fp16 products) against the exact-fp32 MFMA yardstick -- PSNR of the
fine render on the held-out rays every 500 steps.  Chaotic trajectories differ step by step; what is compared is where they end.
    python tools/psnr_long_run.py --steps 5000 --out gpurun_out/psnr_r06.json"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--out", required=True)
    ap.add_argument("--ensemble", type=int, default=0,
                    help="per arithmetic, this many extra runs whose initial weights are moved by a relative 1e-7 (one fp32 rounding "
                         "is 6e-8): the spread of a chaotic trajectory under ITS OWN arithmetic is the yardstick for the other's")
    a = ap.parse_args()
    import psnr_trajectory as T
    rec = {"steps": a.steps, "n_rand": T.N_RAND, "samples": [T.S_C, T.S_F], "lr": T.LR, "scene": "procedural (scnerf_amd/synthetic.py)"}
    for mode in ("fp32", "resident"):
        t0 = time.time()
        curve = T.run_gpu(a.steps, 500, mode)
        rec[mode] = {"curve": curve, "final_psnr": curve[-1]["psnr"], "seconds": time.time() - t0}
        if a.ensemble:
            finals = [T.run_gpu(a.steps, a.steps, mode, perturb=1e-7, perturb_seed=k)[-1]["psnr"] for k in range(a.ensemble)]
            mean = sum(finals) / len(finals)
            rec[mode]["ensemble_final_psnr"] = finals
            rec[mode]["ensemble_mean"] = mean
            rec[mode]["ensemble_std"] = (sum((x - mean) ** 2 for x in finals) / max(len(finals) - 1, 1)) ** 0.5
    rec["final_psnr_difference_db"] = rec["resident"]["final_psnr"] - rec["fp32"]["final_psnr"]
    rec["max_abs_difference_along_the_curve_db"] = max(abs(a_["psnr"] - b_["psnr"]) for a_, b_ in zip(rec["fp32"]["curve"], rec["resident"]["curve"]))
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps({k: rec[k] for k in ("final_psnr_difference_db", "max_abs_difference_along_the_curve_db")}),
          rec["fp32"]["final_psnr"], rec["resident"]["final_psnr"],
          {m: (rec[m].get("ensemble_mean"), rec[m].get("ensemble_std")) for m in ("fp32", "resident")})


if __name__ == "__main__":
    main()
