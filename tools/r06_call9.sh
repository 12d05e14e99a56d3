#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06i; mkdir -p $O
for c in 3 4; do
timeout 300 rocprofv3 --kernel-trace -d $O/trace$c -o b -- python bench.py --config $c --steps 10 --warmup 3 --no-cpu --no-extras --telemetry-seconds 0 --detail $O/d$c.json > $O/bench$c.json 2> $O/trace.err
python tools/rocpd_summary.py $(find $O/trace$c -name "*.db" | head -1) > $O/kernel_trace_config$c.txt
rm -rf $O/trace$c
python -c "import json; d=json.load(open('$O/bench$c.json')); print('config $c (traced)', d['ms_per_step'], d['ms_per_step_events_off'])"
head -3 $O/kernel_trace_config$c.txt
grep -c . $O/kernel_trace_config$c.txt
done
timeout 200 python bench.py --config 4 --steps 10 --warmup 3 --no-cpu --no-extras --detail $O/d4b.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 4', d['ms_per_step'], d['value'])"
sed -n 4,40p $O/kernel_trace_config3.txt | cut -c1-130
