#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
for c in 1 2 3; do timeout 120 python tools/sync_debug.py --config $c >> $O/sync_debug.txt 2>&1; done
cat $O/sync_debug.txt | grep -v "^/opt\|Warning\|warn" | head -40
for c in 2 3; do timeout 200 python bench.py --config $c --steps 10 --warmup 3 --no-cpu --no-extras --detail $O/d$c.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config', d['config']['baseline_config'], d['ms_per_step'], d['ms_per_step_events_off'])"; done
timeout 200 python bench.py --config 3 --prd-sync --steps 10 --warmup 3 --no-cpu --no-extras --detail $O/d3s.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 3 prd-sync', d['ms_per_step'], d['ms_per_step_events_off'])"
(timeout 900 python -m pytest "tests/test_gpu_camera.py::test_combined_config3_step_gradients_with_decisions_aligned" tests/test_gpu_prd.py "tests/test_bench_line.py::test_gpu_bench_measures_its_hbm_traffic_in_the_run" -m gpu -q --timeout 600 2>&1 | tail -5)
