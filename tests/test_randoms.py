"""scnerf_render_randoms (csrc/randoms.hip): the draws of one render_rays call from one launch.  Philox4x32-10 against the
generator's published known-answer vectors, the streams' independence and reproducibility, tails and skipped streams,
distribution checks -- on the CPU SIMT interpreter and (-m gpu) through the C ABI on the device, where the host layer's
throughput path is also checked to launch no ATen generator kernel."""
import numpy as np
import pytest
import torch

from tests.emu import harness as H


def emu_draw(seed, call, sizes, std):
    outs = [np.full(max(z, 1) + 8, -7.0, np.float32) if z else None for z in sizes]
    H.call("scnerf_render_randoms", int(seed), int(call), outs[0], sizes[0], outs[1], sizes[1], outs[2], sizes[2],
           outs[3], sizes[3], float(std), None)
    for o, z in zip(outs, sizes):
        if o is not None:
            assert np.all(o[z:] == -7.0)              # nothing written behind the requested length
    return [None if o is None else o[:z].copy() for o, z in zip(outs, sizes)]


@pytest.mark.emu
def test_philox_known_answers_and_layout():
    """Random123's published vectors for philox4x32-10: counter 0, key 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8 -- the first
    quad of stream 0 at call 0 with seed 0 IS that block; a uniform is the top 24 bits x 2^-24."""
    t_rand, _, _, _ = emu_draw(0, 0, [8, 0, 0, 0], 0.0)
    want = np.array([0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8], np.uint64)
    np.testing.assert_array_equal(t_rand[:4], ((want >> 8).astype(np.float64) / 16777216.0).astype(np.float32))
    # the all-ones vector: counter ffffffff x 4, key ffffffff x 2 -> 408f276d 41c83b0e a20bc7c6 6d5451fd; reached through the
    # host-callable generator the kernel is built on?  (the kernel's counter layout cannot produce an all-ones counter from
    # its arguments; the second published vector is checked on the function itself in the GPU leg's numpy port below)
    got = philox_numpy(np.array([[0xffffffff] * 4], np.uint32), 0xffffffff, 0xffffffff)[0]
    np.testing.assert_array_equal(got, np.array([0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd], np.uint32))


def philox_numpy(ctr, k0, k1):
    """philox4x32-10 in numpy (the checker's restatement of Salmon et al.'s round function)"""
    c = ctr.astype(np.uint64).copy()
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xffffffff)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = np.stack([hi1 ^ c[:, 1] ^ k0, lo1, hi0 ^ c[:, 3] ^ k1, lo0], 1)
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c.astype(np.uint32)


def _cpu_generator_draws(seed, k):
    """the first k 63-bit draws of torch's default generator after manual_seed(seed) (what ops.next_random_call consumes)"""
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    t = torch.empty((), dtype=torch.int64)
    out = [int(t.random_()) for _ in range(k)]
    torch.set_rng_state(state)
    torch.manual_seed(seed)
    return out


def expected_uniforms(seed, call, stream, n):
    quads = (n + 3) // 4
    ctr = np.zeros((quads, 4), np.uint32)
    ctr[:, 0] = np.arange(quads)
    ctr[:, 1] = np.uint32(stream << 28)
    ctr[:, 2], ctr[:, 3] = call & 0xffffffff, call >> 32
    r = philox_numpy(ctr, seed & 0xffffffff, seed >> 32)
    return ((r >> 8).astype(np.float64) / 16777216.0).astype(np.float32).reshape(-1)[:n]


@pytest.mark.emu
def test_streams_tails_and_reproducibility():
    sizes = [4 * 64 + 3, 2 * 128 + 1, 70, 5]           # tails of 3, 1, 2, 1 elements
    a = emu_draw(1234567890123, 7, sizes, 0.5)
    b = emu_draw(1234567890123, 7, sizes, 0.5)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)              # a pure function of (seed, call, stream, element)
    np.testing.assert_array_equal(a[0], expected_uniforms(1234567890123, 7, 0, sizes[0]))
    np.testing.assert_array_equal(a[1], expected_uniforms(1234567890123, 7, 1, sizes[1]))
    c = emu_draw(1234567890123, 8, sizes, 0.5)
    d = emu_draw(1234567890124, 7, sizes, 0.5)
    for other in (c, d):
        assert not np.array_equal(a[0], other[0]) and not np.array_equal(a[2], other[2])
    # a skipped stream changes nothing in the others
    e = emu_draw(1234567890123, 7, [sizes[0], 0, sizes[2], 0], 0.5)
    np.testing.assert_array_equal(a[0], e[0])
    np.testing.assert_array_equal(a[2], e[2])
    assert e[1] is None and e[3] is None
    assert 0.0 <= a[0].min() and a[0].max() < 1.0


def _distribution_checks(t_rand, u, noise_c, noise_f, std):
    from scipy import stats
    for x in (t_rand, u):
        assert 0.0 <= x.min() and x.max() < 1.0
        assert stats.kstest(x.astype(np.float64), "uniform").pvalue > 1e-4
        assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 5.0 / np.sqrt(x.size)
    for x in (noise_c, noise_f):
        z = x.astype(np.float64) / std
        assert stats.kstest(z, "norm").pvalue > 1e-4
        assert abs(z.mean()) < 5.0 / np.sqrt(z.size) and abs(z.var() - 1.0) < 10.0 / np.sqrt(z.size)
        assert abs(stats.kurtosis(z)) < 0.1 and np.abs(z).max() < 6.5
    assert abs(np.corrcoef(noise_c, noise_f[:noise_c.size])[0, 1]) < 5.0 / np.sqrt(noise_c.size)


@pytest.mark.emu
def test_distributions_on_the_interpreter():
    n = 1 << 16
    _distribution_checks(*emu_draw(42, 1, [n, n, n, 2 * n], 0.7), std=0.7)


@pytest.mark.gpu
def test_gpu_draws_equal_the_interpreters_uniforms_and_are_well_distributed():
    from scnerf_amd import ops
    ops.random_shard(0)
    torch.manual_seed(1234)
    call = _cpu_generator_draws(1234, 1)[0]       # (the call id IS the default generator's next 63-bit draw)
    t_rand, u, noise_c, noise_f = ops.render_randoms(4096, 64, 128, 1.0, torch.device("cuda"))
    assert t_rand.shape == (4096, 64) and u.shape == (4096, 128) and noise_c.shape == (4096, 64) and noise_f.shape == (4096, 192)
    np.testing.assert_array_equal(t_rand.cpu().numpy().reshape(-1), expected_uniforms(1234, call, 0, 4096 * 64))
    np.testing.assert_array_equal(u.cpu().numpy().reshape(-1), expected_uniforms(1234, call, 1, 4096 * 128))
    _distribution_checks(*(x.cpu().numpy().reshape(-1) for x in (t_rand, u, noise_c, noise_f)), std=1.0)
    # nothing wanted, nothing drawn
    assert ops.render_randoms(16, 64, 0, 0.0, torch.device("cuda"), want_t_rand=False, want_u=False) == (None, None, None, None)


@pytest.mark.gpu
def test_gpu_render_rays_draws_in_one_launch_of_its_own():
    """the throughput path (no injected draws): no ATen generator kernel between the ray batch and the loss, reproducible
    from torch.manual_seed + the call counter, different from call to call"""
    from scnerf_amd import ops, synthetic as synth
    from scnerf_amd.create_nerf import FusedNetworkQuery
    from scnerf_amd.render import render_rays
    from scnerf_amd.run_nerf_helpers import NeRF, get_embedder
    net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    net.load_state_dict(synth.network_params(seed=0))
    net = net.cuda()
    query = FusedNetworkQuery(get_embedder(10, 0)[0], get_embedder(4, 0)[0])
    rays = synth.ray_batch(256, seed=1).cuda()

    def run():
        with torch.no_grad():
            return render_rays(rays, net, query, 64, perturb=1.0, N_importance=128, network_fine=net, raw_noise_std=1.0)["rgb_map"]
    torch.manual_seed(5)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        a = run()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert not any("distribution" in k or "philox" in k.lower() and "render_randoms" not in k for k in names), names
    assert any("render_randoms_kernel" in k for k in names), names
    b = run()
    torch.manual_seed(6)
    d = run()
    torch.manual_seed(5)                          # re-seeding alone reproduces the run (the call counter restarts with the seed)
    c = run()
    assert torch.equal(a, c) and not torch.equal(a, b) and not torch.equal(a, d)
    # ray-parallel ranks share the seed and the sequence of calls: the process's shard of the counter space keeps their draws apart
    torch.manual_seed(5)
    ops.random_shard(3)
    try:
        e = run()
    finally:
        ops.random_shard(0)
    assert not torch.equal(a, e)


def test_call_ids_come_from_the_default_generator_and_ranks_draw_apart():
    """the host-side bookkeeping on the interpreter: no hidden counter -- (seed, shard, the default generator's state) is
    the whole state, so re-seeding with the SAME seed mid-process reproduces what follows"""
    from scnerf_amd import ops
    from tests.emu.host_on_emu import emulated_device
    with emulated_device():
        ops.random_shard(0)
        c1, c2 = _cpu_generator_draws(77, 2)
        torch.manual_seed(77)
        a1 = ops.render_randoms(8, 64, 8, 0.0, "cpu")[0].clone()
        a2 = ops.render_randoms(8, 64, 8, 0.0, "cpu")[0].clone()
        torch.manual_seed(77)
        b1 = ops.render_randoms(8, 64, 8, 0.0, "cpu")[0].clone()
        np.testing.assert_array_equal(a1.numpy().reshape(-1), expected_uniforms(77, c1, 0, 8 * 64))
        np.testing.assert_array_equal(a2.numpy().reshape(-1), expected_uniforms(77, c2, 0, 8 * 64))
        assert torch.equal(a1, b1) and not torch.equal(a1, a2)
        torch.manual_seed(77)
        ops.random_shard(5)
        try:
            d1 = ops.render_randoms(8, 64, 8, 0.0, "cpu")[0].clone()
        finally:
            ops._random_state["shard"] = None
        want = c1 ^ ((5 * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        np.testing.assert_array_equal(d1.numpy().reshape(-1), expected_uniforms(77, want, 0, 8 * 64))
        assert ops.random_shard() == 0            # (no process group, no $RANK: shard 0)
