#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_render.py tests/test_gpu_large_batch.py -m gpu -q --timeout 900 > $O/tests.txt 2>&1); grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^FAILED" $O/tests.txt | head -5 | cut -c1-150
for rep in 1 2; do
  for lib in tools/ubench/lib_scnerf_store_asm.so scnerf_amd/libscnerf_hip.so; do
    SCNERF_HIP_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu --no-extras --no-pmc --detail $O/d.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib', 'ms/step %.3f (events off %.3f)' % (d['ms_per_step'], d['ms_per_step_events_off']))"
  done
done
