#!/bin/bash
# On the GPU box: extra SQ counters of the resident kernels (one counter per pass, --kernel-trace only).
#   bash tools/pmc_h3.sh r03m "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS ..."
set -u
TAG=${1:-r03m}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT/pmc
export TMPDIR=/tmp
for c in "$@"; do
  REPS=2 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc/$c -o $c -- python tools/profile_kernels.py > /dev/null 2> $OUT/pmc_$c.err
  f=$(find $OUT/pmc/$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/pmc/${c}_counter_collection.csv
  rm -rf $OUT/pmc/$c
done
KEEP_ALL= python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_more.txt
rm -rf $OUT/pmc
