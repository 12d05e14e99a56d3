#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_nerfpp.py tests/test_gpu_prd.py tests/test_gpu_parallel.py "tests/test_gpu_camera.py::test_combined_config3_step_gradients_with_decisions_aligned" -m gpu -q --timeout 600 2>&1 | tail -6)
for c in 3 4; do timeout 200 python bench.py --config $c --steps 10 --warmup 3 --no-cpu --no-extras --detail $O/d$c.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config', d['config']['baseline_config'], d['ms_per_step'], d['value'])"; done
timeout 300 python bench.py --gpus 2 --config 4 --steps 5 --warmup 2 --backend gloo --one-device --no-cpu --detail $O/d4n2.json 2>/dev/null | cut -c1-300
