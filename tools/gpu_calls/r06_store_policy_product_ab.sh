#!/bin/bash
# A/B of the activation stores' cache policy in the PRODUCT: lib_scnerf_store_nt.so (a copy of the product build: nt) vs a build whose store_stream_at emits `sc0 sc1 nt` (reverted since)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
for rep in 1 2 3; do
  for lib in tools/ubench/lib_scnerf_store_nt.so scnerf_amd/libscnerf_hip.so; do
    SCNERF_HIP_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu --no-extras --no-pmc --detail $O/d.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); dd = json.load(open('$O/d.json'))
k = dd['kernels']
print('$lib', 'ms/step %.3f (events off %.3f)' % (d['ms_per_step'], d['ms_per_step_events_off']), ' '.join('%s %.3f' % (n.split('/')[0][:18] + '/' + n.split('P=')[1][:6], v['avg_ms']) for n, v in k.items()), 'clock', d['roofline'].get('clock_ghz'), 'W', d['roofline'].get('socket_power_w'))
"
  done
done
(timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_render.py tests/test_gpu_large_batch.py -m gpu -q --timeout 900 2>&1 | tail -4)
