#!/bin/bash
# A/B of two builds (lib_scnerf_before.so vs the build in tree): parity tests, then three rounds of the bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_large_batch.py "tests/test_gpu_render.py::test_training_step_is_bit_reproducible" -m gpu -q --timeout 600 > $O/tests.txt 2>&1); grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^FAILED" $O/tests.txt | head -5 | cut -c1-150
for rep in 1 2 3; do
  for lib in tools/ubench/lib_scnerf_before.so scnerf_amd/libscnerf_hip.so; do
    SCNERF_HIP_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu --no-extras --no-pmc --detail $O/d.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); dd = json.load(open('$O/d.json')); k = dd['kernels']
print('$lib', 'ms/step %.3f (events off %.3f)' % (d['ms_per_step'], d['ms_per_step_events_off']), ' '.join('%.3f' % v['avg_ms'] for n, v in k.items() if 'wgrad(' in n))"
  done
done
