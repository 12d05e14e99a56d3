"""LAB (GPU): does the 256 MB Infinity Cache absorb a produce -> consume hand-off between two launches?  An in-place scale of a
buffer of S bytes (read S + write S per call, back to back), a write-then-read pair of kernels on a ring of S bytes, and the
same with non-temporal-like streaming (torch's copy): effective bandwidth and socket power against S.

    python tools/mall_probe.py > profiles/r05_mall_probe.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench        # noqa: E402


def main():
    dev = torch.device("cuda:0")
    tel = bench.Telemetry(dev)
    print("# tools/mall_probe.py: bytes touched per call / time, socket power (hwmon), 2 s per row")
    print("# %-44s %10s %10s %8s %8s" % ("pattern", "MB", "TB/s", "W", "GHz"))
    for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096):
        n = mb * (1 << 20) // 4
        x = torch.rand(n, device=dev)
        y = torch.empty_like(x)
        for name, fn, nbytes in (("in-place scale (read S, write S)", lambda: x.mul_(1.0001), 2 * n * 4),
                                 ("write S (fill) then read S (sum)", lambda: (y.fill_(1.5), y.sum()), 2 * n * 4),
                                 ("copy S -> S' then read S' (sum)", lambda: (y.copy_(x), y.sum()), 3 * n * 4)):
            r = tel.sample_while(fn, 2.0)
            if r is None:
                print("# no hwmon")
                return
            print("  %-44s %10d %10.2f %8.0f %8.2f" % (name, mb, nbytes / (r["ms_per_step_during"] * 1e-3) / 1e12,
                                                      r["socket_power_w"], r["clock_ghz"]), flush=True)
        del x, y


if __name__ == "__main__":
    main()
