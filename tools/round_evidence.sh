#!/bin/bash
# end-of-round evidence (run on the GPU box: gpurun -- "bash tools/round_evidence.sh r06"): GPU suite, bench line (+ its
# detail file), kernel trace, counters incl. the HBM traffic the bench line quotes (collected LAST, at the sources the
# line was produced from, so `roofline.traffic` is never stale), the fp32 yardstick, the self-started 2-rank launch path.
# Results under gpurun_out/$TAG; copied into profiles/ with the r05_ prefix by tools/keep_evidence.sh.
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/gpu_tests.txt 2>&1; echo rc=$? >> $O/gpu_tests.txt)
timeout 1500 bash tools/collect_profiles.sh $TAG > $O/collect.log 2>&1
# the counters first, the bench line after them: it picks the fresh traffic file up (same sources => same hash)
cp $O/pmc_traffic_$TAG.json profiles/pmc_traffic_$TAG.json 2> /dev/null
timeout 900 python bench.py --detail $O/bench_detail_n1.json > $O/bench_n1.json 2> $O/bench_n1.err
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --one-device --no-cpu --detail $O/bench_detail_n2_gloo_one_device.json > $O/bench_n2_gloo_one_device.json 2> $O/bench_n2.err
timeout 200 python bench.py --mlp-arithmetic fp32 --wgrad-arithmetic fp32 --steps 10 --warmup 3 --no-cpu --no-extras --detail $O/bench_detail_n1_fp32.json > $O/bench_n1_fp32.json 2> /dev/null
# BASELINE.json's other configurations, one rank and two (gloo on the one device: the launch path and the collective's size)
for c in 2 3 4; do
  timeout 200 python bench.py --config $c --steps 10 --warmup 3 --no-cpu --no-extras --detail $O/bench_detail_n1_config$c.json > $O/bench_n1_config$c.json 2> /dev/null
done
# configs[3] with the match count read on the host every step, as the reference's API returns it (the default leaves it on the device)
timeout 200 python bench.py --config 3 --prd-sync --steps 10 --warmup 3 --no-cpu --no-extras --detail $O/bench_detail_n1_config3_prd_sync.json > $O/bench_n1_config3_prd_sync.json 2> /dev/null
for c in 3 4; do
  timeout 300 python bench.py --gpus 2 --config $c --steps 5 --warmup 2 --backend gloo --one-device --no-cpu --detail $O/bench_detail_n2_config$c.json > $O/bench_n2_config${c}_gloo_one_device.json 2> /dev/null
done
timeout 200 python tools/energy_probe.py > $O/energy_probe.txt 2> /dev/null
timeout 300 bash tools/power_trace.sh > $O/power_trace.txt 2> /dev/null
tail -3 $O/gpu_tests.txt
cat $O/bench_n1.json
