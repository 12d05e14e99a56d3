"""PSNR vs reference (the second half of BASELINE.json's metric): a short training run on the procedural scene on the
GPU, in every arithmetic of the training step, against the CPU oracle's trajectory of the SAME run (identical initial
weights, ray batches and random numbers; tests/golden/psnr_oracle.json, written by `tools/psnr_trajectory.py --side
oracle`, the torch-CPU restatement of the reference path: run_nerf.py:495-506, :600-621, run_nerf_helpers.py:10-11)."""
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "psnr_oracle.json")


def test_scene_is_reproducible():
    """the procedural scene both sides train on: fixed by seeds, targets in [0, 1], not degenerate"""
    from scnerf_amd import synthetic as synth
    rays = synth.procedural_rays(n_views=2, res=8)
    t = synth.procedural_targets(rays, n_quad=128)
    assert rays.shape == (128, 11) and t.shape == (128, 3)
    assert float(t.min()) >= 0.0 and float(t.max()) <= 1.0 and float(t.std()) > 0.02
    assert torch.equal(rays, synth.procedural_rays(n_views=2, res=8))
    rec = json.load(open(GOLDEN))
    assert rec["side"] == "oracle" and rec["curve"][0]["step"] == 0 and rec["final_psnr"] > rec["curve"][0]["psnr"] + 5.0


@pytest.mark.gpu
@pytest.mark.parametrize("arithmetic", ["resident", "half", "split", "fp32"])
def test_psnr_trajectory_tracks_the_oracle(arithmetic):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import psnr_trajectory as T
    from scnerf_amd import ops
    from tests import parity_attribution as PA
    want = {c["step"]: c["psnr"] for c in json.load(open(GOLDEN))["curve"]}
    saved = (ops.mlp_arithmetic(), ops.wgrad_arithmetic())
    try:
        curve = T.run_gpu(100, 25, arithmetic)
    finally:
        ops.mlp_arithmetic(saved[0])
        ops.wgrad_arithmetic(saved[1])
    got = {c["step"]: c["psnr"] for c in curve}
    PA.REPORT["psnr_100_steps/" + arithmetic] = {"gpu": got, "oracle": {k: want[k] for k in got}}
    assert got[100] > got[0] + 3.0                                   # it learns
    for step in (25, 50, 75, 100):
        assert abs(got[step] - want[step]) <= 0.1, (arithmetic, step, got[step], want[step])
