#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
timeout 120 python tools/sync_debug.py --config 4 2>&1 | grep -v "sync_debug_mode\|UserWarning" | head -20
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o b -- python bench.py --config 4 --steps 10 --warmup 3 --no-cpu --no-extras --detail $O/d4.json > $O/bench4.json 2> $O/trace.err
python tools/rocpd_summary.py $(find $O/trace -name "*.db" | head -1) > $O/kernel_trace_config4.txt
rm -rf $O/trace
python -c "import json; d=json.load(open('$O/bench4.json')); print('config 4 (traced)', d['ms_per_step'])"
head -45 $O/kernel_trace_config4.txt | cut -c1-140
