#!/bin/bash
# end-of-round evidence (run on the GPU box: gpurun -- "bash tools/round_evidence.sh"): GPU suite, bench line, kernel trace, counters, the other arithmetics, the 2-rank launch path; results under gpurun_out/r03, copied by hand into profiles/
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
(timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r03/gpu_tests.txt 2>&1; echo rc=$? >> gpurun_out/r03/gpu_tests.txt)
timeout 900 python bench.py > gpurun_out/r03/bench_n1.json 2> gpurun_out/r03/bench_n1.err
timeout 600 bash tools/collect_profiles.sh r03 > gpurun_out/r03/collect.log 2>&1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --one-device --no-cpu > gpurun_out/r03/bench_n2_gloo_one_device.json 2> gpurun_out/r03/bench_n2.err
for a in half split fp32; do
  timeout 200 python bench.py --mlp-arithmetic $a --steps 10 --warmup 3 --no-cpu --no-extras > gpurun_out/r03/bench_n1_mlp_$a.json 2> /dev/null
done
tail -3 gpurun_out/r03/gpu_tests.txt
