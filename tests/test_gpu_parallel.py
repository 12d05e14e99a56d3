"""Ray-parallel step on the GPU: two processes share the one MI355X of the test box ("N processes
on 1 GPU"), each rendering its shard of a 1025-ray batch (513 + 512: unequal on purpose) at 64 + 128 samples with
the weight gradients accumulated straight into the attached flat buffer; the collective is gloo on the device
tensors because RCCL refuses two ranks on one device -- the collective is not what is under test, the flat-buffer
plumbing across ranks is.  Same worker and same checks as the CPU (SIMT interpreter) variant in
tests/test_parallel_cpu.py.
Round 5: two processes on one GPU are also the only setting in which kernels of DIFFERENT launches of this library share a SIMD; until
the one-wave-per-SIMD kernels claimed the whole register file, the camera backward kernel beside the other process's data-gradient
kernel lost a register write now and then and this test failed about once in two hundred runs (profiles/r05_two_process_probe.txt)."""
import numpy as np
import pytest
import torch.multiprocessing as mp

from tests.test_parallel_cpu import _check_ray_parallel, _free_port

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("with_optimizer", [False, True])
def test_ray_parallel_step_two_processes_one_gpu(tmp_path, with_optimizer):
    from tests.parallel_nerf_worker import worker
    world = 2

    mp.spawn(worker, args=(world, _free_port(), str(tmp_path), "cuda:0", 1025, 64, 128, False, with_optimizer), nprocs=world, join=True)
    _check_ray_parallel(tmp_path, world)
    if with_optimizer:
        np.testing.assert_array_equal(np.load(tmp_path / "param0.npy"), np.load(tmp_path / "param1.npy"))
