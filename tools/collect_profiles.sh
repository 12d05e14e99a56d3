#!/bin/bash
# On the GPU box: rocprofv3 evidence for the round's kernels, written under gpurun_out/$TAG (copied to profiles/).
#   bash tools/collect_profiles.sh r02
# 1. kernel trace of the default bench workload (per-kernel durations; --kernel-trace only)
# 2. counters, ONE per pass (--pmc with --kernel-trace only, as the pool requires), of tools/profile_kernels.py
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu --no-extras > $OUT/bench_traced.json 2> $OUT/trace.err
python tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/kernel_trace_bench.txt
rm -rf $OUT/trace
mkdir -p $OUT/pmc
for c in GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES FETCH_SIZE WRITE_SIZE; do
  REPS=2 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc/$c -o $c -- python tools/profile_kernels.py > /dev/null 2> $OUT/pmc_$c.err
  f=$(find $OUT/pmc/$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/pmc/${c}_counter_collection.csv
  rm -rf $OUT/pmc/$c
done
# 3. whole-step HBM traffic: the same two counters over PMC_STEPS plain default steps
mkdir -p $OUT/pmc_step
for c in FETCH_SIZE WRITE_SIZE; do
  PMC_STEPS=3 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_step/$c -o $c -- python tools/profile_steps.py pmc > /dev/null 2> $OUT/pmc_step_$c.err
  f=$(find $OUT/pmc_step/$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/pmc_step/${c}_counter_collection.csv
  rm -rf $OUT/pmc_step/$c
done
python tools/pmc_summary.py $OUT/pmc --json $OUT/pmc_traffic_$TAG.json --steps-dir $OUT/pmc_step --steps 3 > $OUT/pmc_kernels.txt
rm -rf $OUT/pmc $OUT/pmc_step
tail -n +1 $OUT/pmc_kernels.txt | head -80
