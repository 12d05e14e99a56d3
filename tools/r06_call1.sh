#!/bin/bash
# round 6, first GPU call: the stand-alone guest-write reproducer, the GPU suite at the new sources, the bench line, inference
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
timeout 120 tools/ubench/guest_write_lab 6 > $O/guest_write_lab.txt 2>&1; echo "rc=$?" >> $O/guest_write_lab.txt
(timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo rc=$? >> $O/gpu_tests.txt)
timeout 600 python bench.py --detail $O/bench_detail_n1.json > $O/bench_n1.json 2> $O/bench_n1.err
timeout 200 python tools/bench_infer.py --images 6 > $O/bench_infer.json 2>&1
cat $O/guest_write_lab.txt; tail -5 $O/gpu_tests.txt; cat $O/bench_n1.json; cat $O/bench_infer.json
