"""bench.py's stdout contract: ONE line, small enough for the driver's bounded stdout tail (round 3's line had grown to
26 KB and came back unparsed), carrying the contract's keys + `roofline` + `cpu_baseline`; everything else goes to the
detail file.  The GPU leg starts `bench.py --gpus 2` WITHOUT a launcher (the path the driver's command line would take if
it ever omitted torch.distributed.run) and parses what comes out."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "detail")


def _fat_record():
    """a full record as main() assembles it, with every free-text field and table blown up well past what a run produces"""
    prose = "x" * 3000
    kernels = {"kernel %d/P=786432" % i: {"avg_ms": 1.234567891234, "pipe": prose, "tflops": 301.123456789} for i in range(200)}
    return {
        "metric": "rays/sec (64+128 samples/ray) train-step", "value": 326015.123456789, "unit": "rays/s", "n_gpus": 8,
        "steps": 20, "warmup": 5, "ms_per_step": 12.5612345678, "ms_per_step_events_off": 12.56, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32 (3 x fp16 MFMA products per product, fp32 accumulate)",
        "arithmetic": {"a": prose, "b": prose}, "timed_region": prose, "data": "synthetic",
        "config": {"workload": "configs[1]: 4096 rays x (64 coarse + 128 fine), " + "y" * 400, "rays_per_gpu": 4096,
                   "parallelism": "ray-parallel x8, 1 RCCL all-reduce/step of 1202945 floats", "other": prose},
        "roofline": {"bound": "mfma", "kernel": "mlp_fwd_h3_kernel/P=786432/train", "achieved": 300.812345678, "peak": 833.3333333,
                     "unit": "TFLOP/s", "frac": 0.36097, "traffic": 8729500128, "traffic_over_algorithmic": 1.0645,
                     "traffic_over_survey_algorithmic": 396.4, "survey_algorithmic_bytes_per_launch": 22020096,
                     "avg_launch_ms": 3.103, "frac_mfma": 0.36, "frac_hbm": 0.33, "algorithmic_bytes_per_launch": 8200000000,
                     "flop_per_launch": 933400000000, "step_traffic_bytes": 46e9, "step_traffic_over_survey_algorithmic": 2246.1,
                     "peak_note": prose, "measured": prose, "pipe": prose, "traffic_source": prose},
        "kernels": kernels, "step_flop_algorithmic": 3733400000000, "step_tflops": 297.123456,
        "per_rank_ms_per_step": [12.5612345678] * 8,
        "all_reduce_alone": {"ms_per_all_reduce": 0.0512345, "floats": 1202945, "world_size": 8, "backend": "nccl"},
        "extras": {"all_fp32_mfma_step": {"ms_per_step": 28.7123456, "rays_per_s": 142812.3456, "step_tflops_over_fp32_mfma_peak": 0.8312345,
                                          "kernels": kernels},
                   "config2_camera_curriculum": {"states": {"none": {"kernels": kernels}}}, "psnr_vs_reference": {"note": prose}},
        "cpu_baseline": {"value": 287.0412345, "unit": "rays/s", "cores": 16, "kind": "reference", "sample": "z" * 500,
                         "best_ms": 14270.123456, "median_ms": 14400.123456, "value_median": 284.4, "os_cpu_count": 256, "threads": 16,
                         "rays_per_s_1024_rays": 300.1, "rays_per_s_1024_rays_anomaly_on_as_shipped": 250.2,
                         "rays_per_s_1024_rays_all_256_threads": 100.3},
        "speedup_vs_cpu_baseline": 1135.812345,
    }


def test_compact_line_is_small_and_complete():
    import bench
    full = _fat_record()
    assert len(json.dumps(full)) > 100_000
    text = bench.compact_record(full, "profiles/bench_detail_n8.json")
    assert "\n" not in text and len(text) < bench.COMPACT_LIMIT == 4096
    line = json.loads(text)
    for k in REQUIRED:
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["n_gpus"] == 8
    assert set(line["config"]) == {"workload", "rays_per_gpu", "parallelism"}
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
              "traffic_over_survey_algorithmic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert abs(line["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-4
    assert line["all_fp32_mfma_step"]["ms_per_step"] == pytest.approx(28.712, rel=1e-4)
    assert "kernels" not in line and "extras" not in line and "arithmetic" not in line
    assert "3 x fp16" in line["dtype"]


def test_compact_line_survives_missing_optional_parts():
    import bench
    full = _fat_record()
    for k in ("extras", "cpu_baseline", "speedup_vs_cpu_baseline", "per_rank_ms_per_step", "all_reduce_alone"):
        full.pop(k)
    full["roofline"] = None
    line = json.loads(bench.compact_record(full, "x.json"))
    assert line["roofline"] is None and line["cpu_baseline"] is None and line["value"] == full["value"]


def _run_bench(args, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 4096
    return json.loads(lines[0])


@pytest.mark.gpu
def test_gpu_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher: two ranks through torch.distributed.run on 127.0.0.1 (gloo, both on
    the one device of the box: a functional check of the launch path), ONE parseable line from rank 0."""
    detail = str(tmp_path / "detail.json")
    line = _run_bench(["--gpus", "2", "--backend", "gloo", "--one-device", "--steps", "2", "--warmup", "1", "--no-cpu",
                       "--rays", "512", "--detail", detail])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["value"] > 0 and len(line["per_rank_ms_per_step"]) == 2
    assert line["roofline"]["frac"] > 0
    full = json.load(open(detail))
    assert "kernels" in full and full["n_gpus"] == 2


@pytest.mark.gpu
@pytest.mark.parametrize("config,floats", [(1, 2 * 595844), (3, None), (4, None)])
def test_gpu_bench_runs_the_named_configs_on_two_ranks(tmp_path, config, floats):
    """`--config k --gpus 2`: BASELINE.json's configs[1], [3] (learnable camera + projected-ray-distance term every step) and
    [4] (NeRF++) with two ranks on one device over gloo -- the workload is named in the line, the per-rank times and the
    collective on its own are there, and the flat gradient buffer has the size the configuration implies."""
    detail = str(tmp_path / "detail.json")
    line = _run_bench(["--gpus", "2", "--config", str(config), "--backend", "gloo", "--one-device", "--steps", "2", "--warmup", "1",
                       "--no-cpu", "--rays", "256", "--detail", detail])
    assert line["n_gpus"] == 2 and line["config"]["baseline_config"] == config
    assert line["config"]["workload"].startswith("configs[%d]" % config)
    assert "all-reduce/step" in line["config"]["parallelism"]
    assert len(line["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in line["per_rank_ms_per_step"])
    assert "all_reduce_alone" in line
    full = json.load(open(detail))
    n = full["config"]["flat_gradient_floats"]
    if floats is not None:
        assert n == floats
    elif config == 3:
        assert n > 2 * 595844                       # both networks AND the learnable camera tensors
    else:
        assert n == 2 * (595844 + 606596)           # two NerfNet levels x (foreground + background network)


@pytest.mark.gpu
def test_gpu_bench_says_no_collective_on_one_rank(tmp_path):
    line = _run_bench(["--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras", "--rays", "256", "--config", "3",
                       "--telemetry-seconds", "0.3", "--detail", str(tmp_path / "d.json")])
    assert "no collective at N = 1" in line["config"]["parallelism"] and line["config"]["baseline_config"] == 3
    roof = line["roofline"]
    if "clock_ghz" in roof:                         # (hwmon files present: they are on the pool's boxes)
        assert 0.3 < roof["clock_ghz"] < 2.6 and 100 < roof["socket_power_w"] < 1600


@pytest.mark.gpu
def test_gpu_bench_single_line(tmp_path):
    detail = str(tmp_path / "detail.json")
    line = _run_bench(["--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras", "--no-pmc", "--rays", "1024", "--detail", detail])
    for k in REQUIRED:
        assert k in line, k
    assert line["n_gpus"] == 1 and line["cpu_baseline"] is None
    assert line["roofline"]["bound"] in ("mfma", "hbm") and 0 < line["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_gpu_bench_measures_its_hbm_traffic_in_the_run(tmp_path):
    """`roofline.traffic` and `step_traffic_bytes` come from two rocprofv3 counter passes started BY this bench run (one
    counter per pass, --kernel-trace only), not from a committed file: at 512 rays the dominant launch (fine forward or fine
    data gradients, 98 304 samples) must have moved about its workspace (10.4 / 10.1 KB per sample), and a step 40-odd KB per
    sample."""
    detail = str(tmp_path / "detail.json")
    line = _run_bench(["--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras", "--rays", "512", "--detail", detail], timeout=900)
    roof = line["roofline"]
    full = json.load(open(detail))["roofline"]
    assert roof.get("traffic_measured_live") is True, full.get("traffic_live_note") or full.get("traffic_source")
    P = 512 * 192
    assert 0.8 * 10000 * P < roof["traffic"] < 1.5 * 10400 * P, roof
    assert 30e3 * 512 * 256 < roof["step_traffic_bytes"] < 60e3 * 512 * 256, roof
    assert "rocprofv3" in full["traffic_source"] and full["traffic_pass_seconds"] < 300


# ---- first contact at N > 1 must fail LOUDLY, never hang (no GPU needed: the failures happen before any kernel) ----------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_env(rank, world, port):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    return env


def _error_line(stdout):
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "error", "stage"):
        assert k in line, k
    assert line["value"] is None and line["error"]
    return line


def test_missing_peer_ends_in_an_error_line_not_a_hang():
    """rank 0 of a two-rank job whose peer never shows up: the rendezvous deadline passes, ONE JSON line with `error` and the
    stage's name comes out on stdout, the status is non-zero -- in seconds, not at the driver's own timeout"""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--init-timeout", "4",
                        "--steps", "1", "--warmup", "0", "--no-cpu"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=300, env=_rank_env(0, 2, _free_port()), cwd=ROOT)
    assert r.returncode != 0
    line = _error_line(r.stdout)
    assert line["n_gpus"] == 2 and "init_process_group" in line["stage"] and line["rank"] == 0
    assert time.time() - t0 < 240
    assert "FAILED in stage" in r.stderr


def test_sigterm_from_the_launcher_ends_in_an_error_line():
    """what torch.distributed.run does to the surviving ranks when one dies: SIGTERM while rank 0 is blocked inside the
    rendezvous (a C call) -- the sigwait thread answers with the error line and status 4"""
    import signal
    import time
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--init-timeout",
                          "600", "--no-cpu"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         env=_rank_env(0, 2, _free_port()), cwd=ROOT)
    # wait until the process says it is in the rendezvous (stderr is line-buffered through sys.stderr.write + flush)
    t0 = time.time()
    while time.time() - t0 < 240:
        ln = p.stderr.readline()
        if "entering" in ln or p.poll() is not None:
            break
    time.sleep(1.0)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 4, (p.returncode, err[-2000:])
    line = _error_line(out)
    assert "SIGTERM" in line["error"] and "init_process_group" in line["stage"]


def test_two_ranks_without_a_gpu_fail_through_the_self_launcher():
    """`python bench.py --gpus 2` on a machine without a GPU: both ranks meet over gloo, the device step raises, rank 0's
    error line is handed through by the self-launcher, status non-zero"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine WITHOUT a GPU (the failure is the missing device)")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device",
                        "--steps", "1", "--warmup", "0", "--no-cpu", "--init-timeout", "120"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    line = _error_line(r.stdout)
    assert line["n_gpus"] == 2


@pytest.mark.gpu
def test_gpu_two_rccl_ranks_on_one_device_end_with_a_line_not_a_hang():
    """First contact with RCCL at N > 1 as far as a one-GPU box allows: `bench.py --gpus 2 --backend nccl --one-device` -- a
    real rendezvous, a real ncclCommInitRank with world size 2.  RCCL normally refuses two ranks on one device ("Duplicate GPU
    detected"): then rank 0's error line comes out, status non-zero, in seconds.  (Should a build accept it, the run is a
    genuine two-rank all-reduce and must produce the normal line.)  What must never happen is a hang up to the driver's limit."""
    import time
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "nccl", "--one-device",
                        "--init-timeout", "90", "--steps", "2", "--warmup", "1", "--no-cpu", "--rays", "256"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout[-1000:], r.stderr[-3000:])
    line = json.loads(lines[0])
    assert time.time() - t0 < 600
    if r.returncode == 0:
        assert line["n_gpus"] == 2 and line["all_reduce_alone"]["backend"] == "nccl" and line["ranks"]["world_size"] == 2
    else:
        assert line["value"] is None and line["error"] and line["n_gpus"] == 2, line
        assert "init_process_group" in line["stage"] or "collective" in line["stage"] or "warm-up" in line["stage"], line
    print("\ntwo RCCL ranks on one device:", "ran" if r.returncode == 0 else "refused in stage '%s': %s" % (line["stage"], line["error"][:300]))


def test_live_counter_passes_are_parsed_as_the_guide_prescribes(tmp_path, monkeypatch):
    """bench.live_pmc_traffic without a GPU: a stand-in `rocprofv3` on PATH writes the counter CSVs the real tool writes (one
    counter per pass); bytes = WRITE_SIZE KB x 1024 + 2 x FETCH_SIZE KB x 1024, per step over every dispatch, per launch over the
    LARGER launches of the dominant kernel (a kernel that serves both passes under one name is launched at two sizes)."""
    import stat
    import bench
    fake = tmp_path / "rocprofv3"
    fake.write_text('''#!%s
import csv, os, sys
a = sys.argv[1:]
counter, out = a[a.index("--pmc") + 1], a[a.index("-d") + 1]
assert "--kernel-trace" in a and "--" in a and a[a.index("--") + 2].endswith("profile_steps.py")
steps = int(os.environ["PMC_STEPS"])
os.makedirs(os.path.join(out, "host", "123"), exist_ok=True)
rows = []
for s in range(steps):
    # per step: the dominant kernel at two sizes (fine: grid 6144, coarse: grid 2048) + a small kernel
    for name, grid, kb in (("void scn::h3b::mlp_bwd_h3_kernel<3>(float const*)", 1572864, {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 7000.0}),
                           ("void scn::h3b::mlp_bwd_h3_kernel<3>(float const*)", 524288, {"FETCH_SIZE": 300.0, "WRITE_SIZE": 2300.0}),
                           ("vecmat_kernel(float const*)", 65536, {"FETCH_SIZE": 400.0, "WRITE_SIZE": 1.0})):
        rows.append({"Kernel_Name": name, "Grid_Size": grid, "Counter_Name": counter, "Counter_Value": kb[counter], "Dispatch_Id": len(rows)})
with open(os.path.join(out, "host", "123", counter + "_counter_collection.csv"), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0]))
    w.writeheader()
    w.writerows(rows)
''' % sys.executable)
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    got, note = bench.live_pmc_traffic("mlp_bwd_h3_kernel/P=98304", 512, steps=3)
    assert got is not None, note
    assert got["bytes_per_launch"] == int(7000.0 * 1024 + 2 * 1000.0 * 1024)                     # the fine launches only
    assert got["bytes_per_step"] == int((7000 + 2300 + 1) * 1024 + 2 * (1000 + 300 + 400) * 1024)
    assert "two separate passes" in note and "FETCH_SIZE" in note
    # a region the table does not know: the per-step figure is still there, the per-launch one is None
    got2, _ = bench.live_pmc_traffic("some_other_kernel/P=1", 512, steps=3)
    assert got2["bytes_per_launch"] is None and got2["bytes_per_step"] == got["bytes_per_step"]
