#!/bin/bash
# rehearsal of the N-rank launch path as far as one GPU allows: 4 and 8 ranks over gloo on the one device (functional, not a timing)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
for n in 4 8; do
  timeout 600 python bench.py --gpus $n --steps 2 --warmup 1 --backend gloo --one-device --no-cpu --rays 1024 --detail $O/d_n$n.json > $O/bench_n${n}_gloo_one_device.json 2> $O/err_n$n.txt
  echo "n=$n rc=$?"; cut -c1-1500 $O/bench_n${n}_gloo_one_device.json
done
timeout 600 python bench.py --gpus 8 --config 3 --strong --rays 4096 --steps 2 --warmup 1 --backend gloo --one-device --no-cpu --detail $O/d_n8c3.json > $O/bench_n8_config3_strong_gloo_one_device.json 2> $O/err_n8c3.txt
echo "n=8 config 3 strong rc=$?"; cut -c1-900 $O/bench_n8_config3_strong_gloo_one_device.json
