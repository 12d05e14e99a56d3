#!/bin/bash
# round 6, second GPU call: the reproducer in two processes, the trained-weights golden, the GPU suite with the new parity cases
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
timeout 300 bash tools/ubench/guest_write_lab_run.sh 6 > $O/guest_write_lab.txt 2>&1
timeout 900 python tools/train_golden_weights.py --steps 5000 --out $O/trained_nerf.npz > $O/train_golden.log 2>&1
cp $O/trained_nerf.npz tests/golden/trained_nerf.npz
(timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo rc=$? >> $O/gpu_tests.txt)
cp gpurun_out/parity_r06.json $O/parity_r06.json 2>/dev/null
cat $O/guest_write_lab.txt; tail -3 $O/train_golden.log; tail -30 $O/gpu_tests.txt
