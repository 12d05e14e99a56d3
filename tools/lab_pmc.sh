#!/bin/bash
# On the GPU box: per-kernel counters (one per pass) of a lab binary.   bash tools/lab_pmc.sh <tag> <binary> [args...]
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT/pmc
export TMPDIR=/tmp
for c in ${COUNTERS:-GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS}; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc/$c -o $c -- "$@" > $OUT/run_$c.txt 2> $OUT/pmc_$c.err
  f=$(find $OUT/pmc/$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/pmc/${c}_counter_collection.csv
  rm -rf $OUT/pmc/$c
done
KEEP_ALL=1 python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_kernels.txt
rm -rf $OUT/pmc
cat $OUT/pmc_kernels.txt | head -120
