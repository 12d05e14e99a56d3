"""LAB (GPU): what the socket draws -- idle, under a streaming copy at the HBM roof, under the default training step, and
under the step's two kinds of kernels on their own -- from the amdgpu hwmon files (bench.Telemetry), so that the
power-capped regime of profiles/r05_lab_residency.txt has per-byte and per-step energy figures beside it.

    python tools/energy_probe.py > profiles/r05_energy_probe.txt"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                        # noqa: E402
from scnerf_amd import mlp_layout as ML, ops, synthetic as synth    # noqa: E402
from scnerf_amd.parallel import FlatGradAllReduce                   # noqa: E402


def main():
    dev = torch.device("cuda:0")
    tel = bench.Telemetry(dev)
    print("# tools/energy_probe.py: amdgpu hwmon power1_input / freq1_input, median of 25 ms samples over 3 s each")
    if tel.dir is None:
        print("# no hwmon files for this device")
        return
    torch.cuda.synchronize()
    time.sleep(1.0)
    idle = [tel._read("power1_input") * 1e-6 for _ in range(20) if time.sleep(0.05) is None]
    p_idle = sorted(idle)[len(idle) // 2]
    print("%-46s %8.0f W   (cap %.0f W)" % ("idle (context up, nothing running)", p_idle, tel._read("power1_cap") * 1e-6))
    # a streaming copy: 1 GiB read + 1 GiB written per call
    a = torch.rand(1 << 28, device=dev)
    b = torch.empty_like(a)
    r = tel.sample_while(lambda: b.copy_(a), 3.0)
    gbs = 2 * a.numel() * 4 / (r["ms_per_step_during"] * 1e-3) / 1e9
    print("%-46s %8.0f W   %5.2f GHz   %.2f TB/s moved -> %.0f pJ per byte above idle" % (
        "device copy (1 GiB read + 1 GiB written)", r["socket_power_w"], r["clock_ghz"], gbs / 1e3,
        (r["socket_power_w"] - p_idle) / (gbs * 1e9) * 1e12))
    rd = tel.sample_while(lambda: a.sum(), 3.0)
    gbs_r = a.numel() * 4 / (rd["ms_per_step_during"] * 1e-3) / 1e9
    print("%-46s %8.0f W   %5.2f GHz   %.2f TB/s read  -> %.0f pJ per byte above idle" % (
        "reduction over 1 GiB (reads only)", rd["socket_power_w"], rd["clock_ghz"], gbs_r / 1e3,
        (rd["socket_power_w"] - p_idle) / (gbs_r * 1e9) * 1e12))
    del a, b
    # the step and its kernels
    w = bench.build_world(dev, 0, 4096)
    red = FlatGradAllReduce([w["net_c"], w["net_f"]], 1)
    step = bench.fixed_camera_step(w, red)
    r = tel.sample_while(step, 3.0)
    e_step = r["socket_power_w"] * r["ms_per_step_during"] * 1e-3
    print("%-46s %8.0f W   %5.2f GHz   %.3f ms per step = %.1f J per step (%.1f J above idle)" % (
        "default training step, 4096 rays x (64+128)", r["socket_power_w"], r["clock_ghz"], r["ms_per_step_during"], e_step,
        e_step - p_idle * r["ms_per_step_during"] * 1e-3))
    lay = ML.layout(3)
    p = synth.network_params(seed=1)
    flat = torch.cat([p[name].reshape(-1) for name, _ in lay.param_shapes]).float().to(dev)
    wpk, wbk, rw = ops.pack_weights(flat, "fwd"), ops.pack_weights(flat, "bwd"), ops.pack_resident(flat)
    P = 4096 * 192
    g = torch.Generator().manual_seed(5)
    pts = (torch.rand(P, 3, generator=g) * 3 - 1.5).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(4096, 3, generator=g), dim=-1).to(dev)
    d_raw = (torch.randn(P, 4, generator=g) * 1e-3).to(dev)
    save = ops.save_workspace(P, dev)
    mx = ops.ChunkMaxima(P, dev)
    ops.mlp_fwd_resident(pts, vd, 192, wpk, rw, save, maxima=mx)
    grads, _, _ = ops.mlp_bwd_resident(d_raw, pts, vd, 192, wbk, rw, save, maxima=mx)
    fg = torch.zeros(lay.n_params, device=dev)
    for name, fn in (("resident forward, training (P = 786 432)", lambda: ops.mlp_fwd_resident(pts, vd, 192, wpk, rw, save, maxima=mx)),
                     ("resident forward, inference", lambda: ops.mlp_fwd_resident(pts, vd, 192, wpk, rw, None)),
                     ("resident data gradients", lambda: ops.mlp_bwd_resident(d_raw, pts, vd, 192, wbk, rw, save, maxima=mx)),
                     ("weight-gradient group (12 GEMMs + reduction)", lambda: ops.nerf_wgrad(save, grads, d_raw, P, flat_grad=fg, maxima=mx))):
        r = tel.sample_while(fn, 3.0)
        print("%-46s %8.0f W   %5.2f GHz   %.3f ms per launch = %.2f J per launch" % (
            name, r["socket_power_w"], r["clock_ghz"], r["ms_per_step_during"], r["socket_power_w"] * r["ms_per_step_during"] * 1e-3))


if __name__ == "__main__":
    main()
