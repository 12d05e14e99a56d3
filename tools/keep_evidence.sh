#!/bin/bash
# here, after `gpurun -- bash tools/round_evidence.sh r06`: copy what is to be judged from gpurun_out/r06 into profiles/
set -eu
TAG=${1:-r06}
cd "$(dirname "$0")/.."
O=gpurun_out/$TAG
cp gpurun_out/parity_$TAG.json profiles/parity_$TAG.json 2> /dev/null || true
for f in gpu_tests.txt bench_n1.json bench_detail_n1.json bench_n2_gloo_one_device.json bench_n1_fp32.json kernel_trace_bench.txt pmc_kernels.txt bench_n1_config2.json bench_n1_config3.json bench_n1_config4.json bench_n1_config3_prd_sync.json bench_n2_config3_gloo_one_device.json bench_n2_config4_gloo_one_device.json energy_probe.txt power_trace.txt; do
  [ -f $O/$f ] && cp $O/$f profiles/${TAG}_$f
done
[ -f $O/pmc_traffic_$TAG.json ] && cp $O/pmc_traffic_$TAG.json profiles/pmc_traffic_$TAG.json
ls -la profiles | grep $TAG
