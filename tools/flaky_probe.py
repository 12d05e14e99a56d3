import sys, os, tempfile, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch.multiprocessing as mp
from tests.parallel_nerf_worker import worker
from tests.test_parallel_cpu import _free_port
from scnerf_amd import mlp_layout as ML
names = [(n, ML.PARAM_OFFSETS[n]) for n, _ in ML.PARAM_SHAPES]
def where(i):
    net = "coarse" if i < 595844 else ("fine" if i < 2 * 595844 else "camera")
    j = i % 595844 if i < 2 * 595844 else i - 2 * 595844
    if net == "camera": return net, j
    best = max((o, n) for n, o in names if o <= j)
    return net, best[1], j - best[0]
if __name__ == "__main__":
    for it in range(int(sys.argv[1])):
        d = tempfile.mkdtemp()
        mp.spawn(worker, args=(2, _free_port(), d, "cuda:0", 1025, 64, 128, False, False), nprocs=2, join=True)
        f0, full = np.load(os.path.join(d, "flat0.npy")), np.load(os.path.join(d, "full.npy"))
        diff = np.abs(f0 - full); i = int(diff.argmax()); scale = np.abs(full).max()
        bad = np.nonzero(diff > 5e-6 * scale)[0]
        if len(bad) == 0 and it % 25 != 0:
            continue
        print("iter %d: max diff %.3e (%.2e of max) at %s; entries over the bar: %d %s" % (it, diff.max(), diff.max() / scale, where(i), len(bad), [where(int(b)) for b in bad[:12]]), flush=True)
        if len(bad):
            np.save(os.path.join(os.getcwd(), "gpurun_out", "flaky_f0.npy"), f0); np.save(os.path.join(os.getcwd(), "gpurun_out", "flaky_full.npy"), full)
            f1 = np.load(os.path.join(d, "flat1.npy")); print("rank buffers equal:", bool((f0 == f1).all()), flush=True)
            break
